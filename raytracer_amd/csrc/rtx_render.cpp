// placeholder (replaced by the path-tracing host orchestration)
#include "rtx_internal.h"
extern "C" {
void rtx_scene_release_pool(rtx_scene* s) { (void)s; }
#define NOTYET return rtx_fail(RTX_ERR_INVALID, "not implemented yet")
int rtx_film_clear(const rtx_film*, int, int, void*) { NOTYET; }
int rtx_render_accumulate(rtx_scene*, const rtx_camera*, const rtx_render_params*, rtx_photons*, const rtx_film*, rtx_stats*) { NOTYET; }
int rtx_film_resolve(const rtx_film*, const rtx_camera*, float*, unsigned char*, float*, int*, float*, float*) { NOTYET; }
int rtx_render(rtx_scene*, const rtx_camera*, const rtx_render_params*, rtx_photons*, float*, unsigned char*, float*, int*, rtx_stats*) { NOTYET; }
int rtx_render_samples(rtx_scene*, const rtx_camera*, const rtx_render_params*, rtx_photons*, float*, rtx_stats*) { NOTYET; }
int rtx_photon_build(rtx_scene*, const rtx_photon_params*, rtx_photons**, rtx_stats*) { NOTYET; }
void rtx_photons_destroy(rtx_photons*) {}
int rtx_photon_info(const rtx_photons*, uint64_t*, float*) { NOTYET; }
int64_t rtx_photon_download(const rtx_photons*, int, rtx_photon*, size_t) { NOTYET; }
int rtx_photon_set(rtx_photons*, int, const void*, size_t, int) { NOTYET; }
void* rtx_photon_device_ptr(const rtx_photons*, int) { return nullptr; }
int rtx_photon_estimate(rtx_photons*, int, const float*, const float*, size_t, float*) { NOTYET; }
}
