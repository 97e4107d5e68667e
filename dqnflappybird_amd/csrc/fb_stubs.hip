// fb_stubs.hip -- TEMPORARY: entry points not implemented yet fail loudly.
#include "fb_common.h"
extern "C" int fb_qnet_create(int arch, int fc_width, int n_actions, int max_batch, fb_qnet_t *out) { return fb_set_error(FB_ERR_STATE, "fb_qnet_create: not implemented yet"); }
extern "C" int fb_qnet_destroy(fb_qnet_t h) { return fb_set_error(FB_ERR_STATE, "fb_qnet_destroy: not implemented yet"); }
extern "C" int fb_qnet_num_params(fb_qnet_t h, int64_t *n_host) { return fb_set_error(FB_ERR_STATE, "fb_qnet_num_params: not implemented yet"); }
extern "C" int fb_qnet_init_params(fb_qnet_t h, int which, uint64_t seed, void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_init_params: not implemented yet"); }
extern "C" int fb_qnet_load_params(fb_qnet_t h, int which, const float *flat , void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_load_params: not implemented yet"); }
extern "C" int fb_qnet_store_params(fb_qnet_t h, int which, float *flat , void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_store_params: not implemented yet"); }
extern "C" int fb_qnet_get_adam_state(fb_qnet_t h, float *m, float *v, float *beta_pows_host) { return fb_set_error(FB_ERR_STATE, "fb_qnet_get_adam_state: not implemented yet"); }
extern "C" int fb_qnet_set_adam_state(fb_qnet_t h, const float *m, const float *v, const float *beta_pows_host) { return fb_set_error(FB_ERR_STATE, "fb_qnet_set_adam_state: not implemented yet"); }
extern "C" int fb_qnet_set_hparams(fb_qnet_t h, float lr, float beta1, float beta2, float eps) { return fb_set_error(FB_ERR_STATE, "fb_qnet_set_hparams: not implemented yet"); }
extern "C" int fb_qnet_forward(fb_qnet_t h, int which, const uint8_t *states, int batch, float *q, void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_forward: not implemented yet"); }
extern "C" int fb_qnet_act(fb_qnet_t h, const uint8_t *states, int n, float epsilon, uint64_t seed, uint64_t step, uint8_t *actions, float *q, void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_act: not implemented yet"); }
extern "C" int fb_qnet_train_step(fb_qnet_t h, int algo, int batch, const uint8_t *s, const uint8_t *a, const float *r, const uint8_t *s2, const uint8_t *t, const float *isw, double gamma, float *loss, float *abs_err, float *q_target, float *flat_grad, void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_train_step: not implemented yet"); }
extern "C" int fb_qnet_apply_adam(fb_qnet_t h, const float *flat_grad, void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_apply_adam: not implemented yet"); }
extern "C" int fb_qnet_sync_target(fb_qnet_t h, void *stream) { return fb_set_error(FB_ERR_STATE, "fb_qnet_sync_target: not implemented yet"); }
