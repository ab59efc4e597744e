// render_fused.hip -- placeholder entry points (replaced by the fused renderer).
#include "ngp_common.hpp"
using namespace ngp;
extern "C" {
int ngp_render_ctx_create(uint32_t max_rays, ngp_render_ctx** out) { (void)max_rays; (void)out; set_error("render_ctx_create: not built"); return NGP_EINVAL; }
int ngp_render_ctx_destroy(ngp_render_ctx* ctx) { (void)ctx; return NGP_OK; }
int ngp_render_rays(ngp_render_ctx* ctx, const ngp_model* model, const float* rays_o, const float* rays_d, const float* nears,
                    const float* fars, uint32_t N, float dt_gamma, uint32_t max_steps, uint32_t perturb, float* weights_sum, float* depth,
                    float* image, ngp_render_stats* stats_host, int sync, ngp_stream_t stream) {
    (void)ctx; (void)model; (void)rays_o; (void)rays_d; (void)nears; (void)fars; (void)N; (void)dt_gamma; (void)max_steps; (void)perturb;
    (void)weights_sum; (void)depth; (void)image; (void)stats_host; (void)sync; (void)stream;
    set_error("render_rays: not built");
    return NGP_EINVAL;
}
int ngp_network_forward(const ngp_model* model, const float* xyzs, const float* dirs, uint32_t M, float* sigmas, float* rgbs,
                        ngp_stream_t stream) {
    (void)model; (void)xyzs; (void)dirs; (void)M; (void)sigmas; (void)rgbs; (void)stream;
    set_error("network_forward: not built");
    return NGP_EINVAL;
}
}
