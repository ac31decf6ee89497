// Mirror of the reference's built-in denoiser entry point (RayTracer/Denoiser.h, Denoiser.cpp:93-134):
// Denoiser::ApplyMedian runs the 3x3 median on the GPU (rtx_median_denoise). OIDN stays external.
#pragma once
#include <string>
#include <vector>
#include "scene.h"
#include "ImageIO.h"
#include "../../include/rtx.h"

namespace rtamd {
namespace Denoiser {

inline bool IsOIDNAvailable() { return false; }

// outputPath empty: no file; outputPixels null: no copy (as the reference)
inline bool ApplyMedian(RenderImage& image, bool outputSRGB, const std::string& outputPath, Color24* outputPixels,
                        std::string* errMsg = nullptr) {
    const int W = image.GetWidth(), H = image.GetHeight();
    if (W <= 0 || H <= 0) {
        if (errMsg) *errMsg = "Invalid image dimensions for median filter.";
        return false;
    }
    static_assert(sizeof(Color) == 12 && sizeof(Color24) == 3, "tightly packed pixels expected");
    std::vector<Color24> out((size_t)W * H);
    if (rtx_median_denoise(&image.GetLinearPixels()->r, W, H, outputSRGB ? 1 : 0, nullptr, &out.data()->r) != RTX_OK) {
        if (errMsg) *errMsg = rtx_last_error();
        return false;
    }
    if (outputPixels) std::copy(out.begin(), out.end(), outputPixels);
    if (!outputPath.empty() && !png::save(outputPath.c_str(), &out.data()->r, W, H, 3)) {
        if (errMsg) *errMsg = "cannot write " + outputPath;
        return false;
    }
    return true;
}

}  // namespace Denoiser
}  // namespace rtamd
