// fb_dist.hip -- the data-parallel step with its all-reduce issued from the C side, straight onto HIP streams this library owns.
//
// One process per GPU; envs and replay are rank-local, the only exchange is the all-reduce of the flat fp32 gradient (3.59 MB) between
// the backward pass and Adam.  Through torch.distributed that collective sits on the critical path and costs two cross-stream hops on
// top (step stream -> RCCL's stream -> back).  Here RCCL is called directly (its symbols come from the librccl the process already
// holds -- dlopen, no link-time dependency) and the vector goes in two pieces:
//   tail   flat_grad[CONV_PARAMS ..): W_fc1, b_fc1, the head -- 91 % of the bytes, final right behind the fc1 backward launch
//          (fb_qnet_set_grad_event): reduced on a SIDE stream that waits for that event, while the conv backward still runs
//   front  flat_grad[.. CONV_PARAMS): the conv layers, final after the slab reduction: reduced on the step's own stream (no hop at all)
// then the step's stream waits for the side stream's completion event and Adam runs.  Both pieces go through ONE communicator in the
// same order on every rank (RCCL serialises them), element for element the same sums as one all-reduce of the whole vector, so the
// replicas stay bit-identical.  Mean losses (BrainDQNNature.py:119) divide by the world size afterwards, like dist.allreduce_gradients.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "fb_common.h"

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} rccl;

int load_rccl(const char *path) {
    if (rccl.lib) return FB_OK;
    const char *tries[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *l = nullptr;
    for (const char *t : tries) {
        if (!t || !*t) continue;
        l = dlopen(t, RTLD_NOW | RTLD_GLOBAL);
        if (l) break;
    }
    FB_REQUIRE(l, "fb_dist: librccl.so could not be loaded (%s)", dlerror());
#define FB_SYM(field, name) rccl.field = (decltype(rccl.field))dlsym(l, name); FB_REQUIRE(rccl.field, "fb_dist: librccl has no %s", name)
    FB_SYM(GetUniqueId, "ncclGetUniqueId");
    FB_SYM(CommInitRank, "ncclCommInitRank");
    FB_SYM(CommDestroy, "ncclCommDestroy");
    FB_SYM(AllReduce, "ncclAllReduce");
    FB_SYM(GetErrorString, "ncclGetErrorString");
#undef FB_SYM
    rccl.lib = l;
    return FB_OK;
}

#define FB_CHECK_NCCL(call)                                                                                  \
    do {                                                                                                     \
        ncclResult_t e_ = (call);                                                                            \
        if (e_ != ncclSuccess) return fb_set_error(FB_ERR_HIP, "%s: %s", #call, rccl.GetErrorString(e_));    \
    } while (0)

__global__ void div_kernel(float *__restrict__ x, long long n, float d) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = x[i] / d;
}
}  // namespace

struct fb_dist {
    ncclComm_t comm;
    int rank, world;
    hipStream_t side;
    hipEvent_t grad_ready, tail_done;
    int overlap;                     // 0 (default): one all-reduce of the whole vector on the step's stream; 1: the two-piece schedule above
};

// rank-local and non-collective: can this process load RCCL at all?  Every rank calls it BEFORE the ranks agree on the native path
// (dist.NativeDP.agree): what can fail alone must fail before anything collective starts.
extern "C" int fb_dist_probe(const char *librccl_path) { return load_rccl(librccl_path); }

extern "C" int fb_dist_unique_id(const char *librccl_path, uint8_t *id128) {
    FB_REQUIRE(id128, "fb_dist_unique_id: NULL argument");
    int rc = load_rccl(librccl_path);
    if (rc != FB_OK) return rc;
    ncclUniqueId id;
    FB_CHECK_NCCL(rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return FB_OK;
}

extern "C" fb_dist_t fb_dist_create(const char *librccl_path, int rank, int world, const uint8_t *id128) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) { fb_set_error(FB_ERR_INVALID, "fb_dist_create: bad argument"); return nullptr; }
    if (load_rccl(librccl_path) != FB_OK) return nullptr;
    fb_dist *d = new fb_dist();
    d->rank = rank; d->world = world; d->overlap = 0;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t e = rccl.CommInitRank(&d->comm, world, id, rank);
    if (e != ncclSuccess) { fb_set_error(FB_ERR_HIP, "ncclCommInitRank: %s", rccl.GetErrorString(e)); delete d; return nullptr; }
    if (hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d->grad_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->tail_done, hipEventDisableTiming) != hipSuccess) {
        fb_set_error(FB_ERR_HIP, "fb_dist_create: stream / event creation failed");
        rccl.CommDestroy(d->comm); delete d; return nullptr;
    }
    return d;
}

extern "C" void fb_dist_destroy(fb_dist_t d) {
    if (!d) return;
    (void)hipStreamSynchronize(d->side);
    rccl.CommDestroy(d->comm);
    (void)hipEventDestroy(d->grad_ready); (void)hipEventDestroy(d->tail_done); (void)hipStreamDestroy(d->side);
    delete d;
}

// all-reduce of the flat gradient, then Adam.  overlap = 1: the two-piece schedule (the producer recorded d->grad_ready behind its fc1
// backward launch); overlap = 0: the whole vector on the step's stream -- no event, no second stream.  Which one wins depends on the
// machine: on MI355X every cross-stream dependency costs ~5-8 us (measured at world size 1: the two-piece schedule adds 21 us of hops
// to a 134 us step, the in-line call 2), so the overlap only pays when the 3.3 MB all-reduce takes longer than that.
static int reduce_and_apply(fb_dist_t d, fb_qnet_t net, float *g, int mean, hipStream_t st) {
    int64_t n = 0;
    int rc0 = fb_qnet_num_params(net, &n);
    if (rc0 != FB_OK) return rc0;
    const long long split = fb_qnet_grad_split(net);
    if (!d->overlap) {
        FB_CHECK_NCCL(rccl.AllReduce(g, g, (size_t)n, ncclFloat, ncclSum, d->comm, st));
        if (mean && d->world > 1) hipLaunchKernelGGL(div_kernel, dim3(256), dim3(256), 0, st, g, (long long)n, (float)d->world);
        return fb_qnet_apply_adam(net, g, st);
    }
    // d->grad_ready must have been recorded by THIS step (behind its fc1 backward launch).  A step that ran without the event installed
    // on the net (fb_qnet_set_grad_event forgotten before fb_qnet_train_step / fb_train_from_replay) has not recorded it, and the side
    // stream would wait on a stale, long completed event and reduce the tail while the backward pass still writes it.  Recording it here
    // in that case is always safe -- it only gives the overlap away for this step.
    if (fb_qnet_take_grad_event_recorded(net) != (void *)d->grad_ready) FB_CHECK_HIP(hipEventRecord(d->grad_ready, st));
    FB_CHECK_HIP(hipStreamWaitEvent(d->side, d->grad_ready, 0));
    FB_CHECK_NCCL(rccl.AllReduce(g + split, g + split, (size_t)(n - split), ncclFloat, ncclSum, d->comm, d->side));
    if (mean && d->world > 1) hipLaunchKernelGGL(div_kernel, dim3(256), dim3(256), 0, d->side, g + split, n - split, (float)d->world);
    FB_CHECK_HIP(hipEventRecord(d->tail_done, d->side));
    FB_CHECK_NCCL(rccl.AllReduce(g, g, (size_t)split, ncclFloat, ncclSum, d->comm, st));
    if (mean && d->world > 1) hipLaunchKernelGGL(div_kernel, dim3(64), dim3(256), 0, st, g, split, (float)d->world);
    FB_CHECK_HIP(hipStreamWaitEvent(st, d->tail_done, 0));
    return fb_qnet_apply_adam(net, g, st);
}

extern "C" int fb_dist_set_overlap(fb_dist_t d, int overlap) {
    FB_REQUIRE(d, "fb_dist_set_overlap: NULL handle");
    d->overlap = overlap != 0;
    return FB_OK;
}

extern "C" int fb_vec_step_dp(fb_dist_t d, fb_env_t env, fb_replay_t replay, fb_qnet_t net, const fb_step_buffers *b, int n_envs, int algo,
                              int batch, float epsilon, uint64_t seed, uint64_t step, int train, double gamma, int mean_loss, void *stream) {
    FB_REQUIRE(d && b && (!train || b->flat_grad), "fb_vec_step_dp: NULL handle / flat_grad buffer");
    void *prev = fb_qnet_get_grad_event(net);          // (an event the caller installed for its own schedule is put back afterwards)
    int rc = fb_qnet_set_grad_event(net, d->overlap ? d->grad_ready : nullptr);
    if (rc != FB_OK) return rc;
    rc = fb_vec_step(env, replay, net, b, n_envs, algo, batch, epsilon, seed, step, train, gamma, stream);
    (void)fb_qnet_set_grad_event(net, prev);
    if (rc != FB_OK || !train) return rc;
    return reduce_and_apply(d, net, b->flat_grad, mean_loss, fb_stream(stream));
}

// the same for a gradient produced by fb_qnet_train_step / fb_train_from_replay with fb_dist_grad_event() set on the net beforehand
extern "C" int fb_dist_reduce_apply(fb_dist_t d, fb_qnet_t net, float *flat_grad, int mean_loss, void *stream) {
    FB_REQUIRE(d && net && flat_grad, "fb_dist_reduce_apply: NULL argument");
    return reduce_and_apply(d, net, flat_grad, mean_loss, fb_stream(stream));
}

extern "C" void *fb_dist_grad_event(fb_dist_t d) { return d ? (void *)d->grad_ready : nullptr; }

// the collective ALONE (sum, in place, on the caller's stream): what bench.py brackets with HIP events to report how long the all-reduce
// itself takes at this world size (config.allreduce_us), apart from the step around it
extern "C" int fb_dist_all_reduce(fb_dist_t d, float *buf, int64_t count, void *stream) {
    FB_REQUIRE(d && buf && count > 0, "fb_dist_all_reduce: bad argument");
    FB_CHECK_NCCL(rccl.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, d->comm, fb_stream(stream)));
    return FB_OK;
}
