// Decodes a texture file with the host mirror's readers and writes its pixels as raw RGB (uint8 or float32) for
// tests/test_loader_cpu.py; prints {"w","h","float"} (host code only).
#include <cstdio>
#include "../../raytracer_amd/host/SceneLoader.h"
using namespace rtamd;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::shared_ptr<TextureMap> t = LoadPnm(argv[1]);
    if (!t) t = imgdec::LoadPng(argv[1]);
    if (!t) t = imgdec::LoadHdr(argv[1]);
    if (!t) {
        std::printf("{\"error\": \"not decoded\"}\n");
        return 1;
    }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 3;
    if (t->isFloat) std::fwrite(t->rgb32f.data(), 4, t->rgb32f.size(), f);
    else std::fwrite(t->rgb8.data(), 1, t->rgb8.size(), f);
    std::fclose(f);
    std::printf("{\"w\": %d, \"h\": %d, \"float\": %s}\n", t->width, t->height, t->isFloat ? "true" : "false");
    return 0;
}
