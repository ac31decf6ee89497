// Mirror of the animation frame loop of the reference's RayTracer/Main.cpp:381-401 (and of its output naming, :272-303):
// for every frame the camera is evaluated at t = start + frame / fps (clamped to `end`), the frame is rendered through
// RayCaster::CastRays and the images are saved. The scene stays resident on the GPU between frames (the GpuScene
// registry of SceneAdapter.h is keyed by the Scene, which the loop does not touch); only the camera changes.
#pragma once
#include <chrono>
#include <functional>
#include <string>
#include <sys/stat.h>
#include "RayCaster.h"

namespace rtamd {

struct SequenceOptions {
    int frames = 1;
    double fps = 24.0, start = 0.0, end = 0.0;   // --anim-frames / --anim-fps / --anim-start / --anim-end
    std::string prefix = "render";
    std::string outDir;                          // empty: nothing is written; else beauty/, zdepth/, samples/ below it
};

struct OutputNames {
    std::string color, z, samples;
};
// Main.cpp:272-303 (sequence branch): <prefix>_<zero-padded frame>.png in beauty/, zdepth/, samples/
inline OutputNames FormatBaseNames(const std::string& dir, const std::string& prefix, int frameIdx, int pad) {
    std::string num = std::to_string(frameIdx);
    if ((int)num.size() < pad) num = std::string((size_t)pad - num.size(), '0') + num;
    const std::string frameName = prefix + "_" + num;
    const std::string root = dir.empty() ? std::string() : dir + "/";
    return {root + "beauty/" + frameName + ".png", root + "zdepth/" + frameName + "_z.png", root + "samples/" + frameName + "_samples.png"};
}

// Returns the number of frames rendered. onFrame(frame, t, image, milliseconds) runs after each frame.
inline int RenderSequence(RayCaster& rc, Camera& camera, RenderImage& image, PhotonMapBuilder* photons, const SequenceOptions& o,
                          const std::function<void(int, double, RenderImage&, double)>& onFrame = {}) {
    const int pad = std::max(4, (int)std::to_string(std::max(o.frames, 1)).size());   // Main.cpp:382
    if (!o.outDir.empty())
        for (const char* sub : {"", "/beauty", "/zdepth", "/samples"}) ::mkdir((o.outDir + sub).c_str(), 0755);
    int done = 0;
    for (int frame = 0; frame < o.frames && !rc.IsStopped(); ++frame) {
        double t = o.start + double(frame) / o.fps;
        if (o.end > o.start && t > o.end) t = o.end;
        camera.EvaluateAtTime(t);
        image.Init(camera.imgWidth, camera.imgHeight);
        const auto t0 = std::chrono::steady_clock::now();
        rc.CastRays(image, photons);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (!o.outDir.empty()) {
            const OutputNames n = FormatBaseNames(o.outDir, o.prefix, frame, pad);
            image.ComputeZBufferImage();
            image.SaveImage(n.color.c_str());
            image.SaveZImage(n.z.c_str());
            image.ComputeSampleCountImage();
            image.SaveSampleCountImage(n.samples.c_str());
        }
        if (onFrame) onFrame(frame, t, image, ms);
        ++done;
    }
    return done;
}

}  // namespace rtamd
