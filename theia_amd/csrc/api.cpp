// C ABI of libtheia_hip.so (include/theia_hip.h): contexts, device memory,
// scene upload, tracer launch/reduce/readback. No torch types, no CPU fallback:
// every entry point fails with THX_ERR_NO_DEVICE / THX_ERR_HIP when there is no
// usable GPU.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "kernels.h"

using namespace thx;

namespace {
thread_local std::string g_error;

// Batches side by side (thx_tracer_set_concurrency) need hardware queues: the HIP runtime multiplexes its streams onto
// GPU_MAX_HW_QUEUES of them (default 4), and streams that share a queue run one after the other (profiles/r04_batches_side_by_side.txt:
// 20 batches of 10^6 photons take 0.31-0.32 ms each with 2 or 4 queues, 0.28 with 8, 0.26 with 16, 0.43 one at a time). The variable is
// read when HIP initialises, so it is set -- unless the user has set it -- when the library is loaded, which is before its first HIP
// call. A process that initialised HIP earlier keeps its setting; thx_ctx_concurrent_kernels measures what it got.
struct HwQueues {
    HwQueues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} g_hwQueues;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}
int hipFail(hipError_t e, const char* what) {
    return fail(THX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define THX_HIP(call)                                        \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hipFail(e_, #call);     \
    } while (0)
} // namespace

struct thx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;          // lane 0
    hipStream_t lanes[THX_MAX_SLOTS] = {}; // lanes[0] == stream; the others are created on first use (laneStream)
    hipStream_t copyStream = nullptr; // readback of one config's hit queue while the other config's batch runs (thx_tracer_read_hits_cfg)
    uint32_t nLanes = 1;
    uint32_t concurrentKernels = 0; // kernels of different streams the runtime runs side by side (0: not measured yet)
    hipDeviceProp_t prop{};
    std::mutex lock; // serialises calls on this context's tracers / scenes (launch bookkeeping, event ring)
};

// Stream of lane `l` of a context: batches in flight side by side (thx_tracer_set_concurrency) run on one stream each.
static hipError_t laneStream(thx_ctx* ctx, uint32_t l, hipStream_t* out) {
    if (l >= THX_MAX_SLOTS) return hipErrorInvalidValue;
    if (!ctx->lanes[l]) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->lanes[l], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        ctx->nLanes = std::max(ctx->nLanes, l + 1u);
    }
    *out = ctx->lanes[l];
    return hipSuccess;
}

struct thx_scene {
    thx_ctx* ctx = nullptr;
    SceneView view{};
    thx_scene_info info{};
    uint32_t nPairs = 0; // pair records the walk reads (inner nodes of all trees)
    bool wide = false;   // ... as 4-wide records of two slots each (bvh_build.cpp)
    float lastQueryMs = 0.0f;
    std::vector<void*> allocs;
};

struct thx_tracer {
    thx_ctx* ctx = nullptr;
    thx_scene* scene = nullptr;
    thx_tracer_desc desc{};
    TraceVariant variant{};
    bool debug = false;
    bool staged = false;          // staged (regrouped-in-LDS) scene kernel
    uint32_t mode = 0;            // bit 0 instrumented, bit 1 staged
    uint32_t stackDepth = kStackDepth;
    uint32_t stackLds = kStackDepth; // entries of the traversal stack kept in LDS; the rest of stackDepth spills to `spill`
    uint32_t stackSlots = kStackDepth; // staged kernel: LDS slots per lane (stackLds, + 1 for the variants with the whole stack in LDS)
    uint32_t* spill = nullptr;
    uint32_t* park = nullptr;     // staged kernel: state of walks set aside half way (dev_walk.hpp: Walker::save)
    bool stack16 = false;         // 16-bit stack entries (small scenes)
    uint32_t pool = 0;            // staged kernel: photons per workgroup pool
    bool exactRings = false;      // staged kernel: ring capacity = pool rounded up to 64 (else: to a power of two)
    uint32_t ringCap = 0;         // staged kernel: cells per ring
    size_t ldsNodeBytes = 0;      // staged kernel: LDS copy of the scene's pair records (0: the walk reads them through the vector cache)
    size_t ldsNodeOffset = 0;     // ... and its byte offset in the workgroup's LDS
    uint32_t block = kTraceBlock;
    size_t lds = 0;
    uint32_t gridMax = 1;
    uint32_t refillBelow = 63;
    // per slot (0 / 1 = the pipeline configs; up to THX_MAX_SLOTS batches in flight): partial histograms, reduced curve,
    // pinned host copy {curve[n_bins], device error word} of an asynchronous readback and the event behind it
    float* slabs[THX_MAX_SLOTS] = {};
    float* out[THX_MAX_SLOTS] = {};
    float* pinned[THX_MAX_SLOTS] = {};
    float* pinnedDev[THX_MAX_SLOTS] = {}; // the same buffers as the device addresses them (the reducer writes them directly)
    bool mirrored[THX_MAX_SLOTS] = {};    // the slot's last reduction went into its pinned buffer as well
    hipEvent_t evFetch[THX_MAX_SLOTS] = {};
    bool fetched[THX_MAX_SLOTS] = {};
    uint32_t lastGrid[THX_MAX_SLOTS] = {};
    uint32_t conc = 1;            // lanes: launches side by side (thx_tracer_set_concurrency), slot s on lane s % conc
    uint32_t gridDiv = 1;         // ... each on at most ceil(gridMax / gridDiv) workgroups
    uint32_t scratchWgs = 0;      // workgroups the scratch arrays below are allocated for (>= conc * laneGrid())
    uint32_t laneGrid() const { return std::max(1u, (gridMax + gridDiv - 1u) / gridDiv); }
    float* cold = nullptr;        // staged kernel: cold chunks of the photon pools in global memory (one array per workgroup)
    size_t coldWg = 0, spillWg = 0, parkWg = 0; // words per workgroup of cold / spill / park
    unsigned long long* stats = nullptr;
    uint32_t* devError = nullptr; // device error word (THX_DEVERR_* bits), set by the kernels' exit paths
    unsigned long long* prof = nullptr;
    float* track = nullptr;
    uint32_t* hits = nullptr;     // HitRecorder queue (count + SoA columns) or null; of pipeline config 0 for the tracing kinds
    uint32_t* hits1 = nullptr;    // ... of pipeline config 1 (tracing kinds with a record / store-time response: a batch of one config
                                  // runs while the other config's queue is read, thx_tracer_read_hits_cfg)
    hipEvent_t evHits[2] = {};    // behind the last launch of either config
    uint32_t lastHitCfg = 0;      // config of the most recent launch (what thx_tracer_read_hits reads)
    uint32_t* replayIn = nullptr; // THX_TRACER_HIT_REPLAY: the input hit queue
    uint32_t* lamOut = nullptr;   // THX_TRACER_LIGHT_SAMPLER: wavelength output queue (light output: `hits`)
    float* lightIn[2] = {nullptr, nullptr}; // HostLightSource queue per pipeline config
    float* lamIn[2] = {nullptr, nullptr};   // HostWavelengthSource queue per pipeline config
    // Cherenkov-family light source of a tracing kind: the sampler kernel fills these right before every launch (one pair per
    // lane: the batches of a lane run one after the other): count word + 8 / 15 columns, count word + wavelength, contrib, dimension
    uint32_t* preLight[THX_MAX_SLOTS] = {};
    uint32_t* preLam[THX_MAX_SLOTS] = {};
    uint32_t trackEvents = 0;
    uint32_t lastBatch = 0;
    // ring of HIP event pairs bracketing the tracing kernel on its launch stream
    static constexpr uint32_t kRing = 64;
    hipEvent_t evStart[kRing] = {}, evStop[kRing] = {};
    uint64_t launches = 0;       // total launches so far
    int lastSlot[THX_MAX_SLOTS];  // ring slot of the most recent launch per slot (-1: none)
    thx_tracer() { std::fill(std::begin(lastSlot), std::end(lastSlot), -1); }
};

// 4-byte columns of a tracer's hit queue (record: response.queue.glsl:6-33, store time:
// response.time.store.glsl:4-10)
static size_t hitColumns(const thx_tracer_desc& d) {
    if (d.kind == THX_TRACER_LIGHT_SAMPLER) return (d.flags & THX_FLAG_POLARIZED) ? 15 : 8; // lightsource.queue.glsl:10-30
    if (d.response_kind == THX_RESPONSE_STORE_TIME) return d.hit_polarized ? 2 : 1;
    return d.hit_polarized ? 20 : 13;
}

// Scratch memory of the staged kernel's workgroups (cold photon chunks, deep stack entries, walks set aside) for `wgs` workgroups
static hipError_t allocScratch(thx_tracer* t, uint32_t wgs) {
    if (wgs <= t->scratchWgs) return hipSuccess;
    for (void* p : {(void*)t->cold, (void*)t->spill, (void*)t->park})
        if (p) (void)hipFree(p);
    t->cold = nullptr; t->spill = nullptr; t->park = nullptr;
    t->scratchWgs = 0;
    hipError_t e = hipSuccess;
    if (t->coldWg) e = hipMalloc((void**)&t->cold, 4 * t->coldWg * wgs);
    if (e == hipSuccess && t->spillWg) e = hipMalloc((void**)&t->spill, 4 * t->spillWg * wgs);
    if (e == hipSuccess && t->parkWg) e = hipMalloc((void**)&t->park, 4 * t->parkWg * wgs);
    if (e == hipSuccess) t->scratchWgs = wgs;
    return e;
}

// Device and pinned host buffers of result slot `slot` (0 and 1 exist from creation on; the memsets run on the null
// stream, which is not ordered with the non-blocking lane streams: they are completed before returning)
static int ensureSlot(thx_tracer* t, uint32_t slot) {
    if (slot >= THX_MAX_SLOTS) return fail(THX_ERR_INVALID, "slot index out of range (THX_MAX_SLOTS)");
    if (t->slabs[slot]) return THX_OK;
    const size_t slabBytes = sizeof(float) * (size_t)t->gridMax * t->desc.n_bins;
    const size_t outBytes = sizeof(float) * (t->desc.n_bins + 1u); // the curve + the device error word as of its reduction
    THX_HIP(hipMalloc((void**)&t->slabs[slot], slabBytes));
    THX_HIP(hipMemset(t->slabs[slot], 0, slabBytes));
    THX_HIP(hipMalloc((void**)&t->out[slot], outBytes));
    THX_HIP(hipMemset(t->out[slot], 0, outBytes));
    THX_HIP(hipHostMalloc((void**)&t->pinned[slot], outBytes + 16, hipHostMallocMapped));
    std::memset(t->pinned[slot], 0, outBytes + 16);
    THX_HIP(hipHostGetDevicePointer((void**)&t->pinnedDev[slot], t->pinned[slot], 0));
    THX_HIP(hipEventCreateWithFlags(&t->evFetch[slot], hipEventDisableTiming));
    THX_HIP(hipDeviceSynchronize());
    return THX_OK;
}

// Reads (and clears) the tracer's device error word. The caller has synchronised the device.
// A tripped scheduler watchdog must never look like a result: every readback goes through here.
static int deviceErrorCheck(thx_tracer* t) {
    if (!t->devError) return THX_OK;
    uint32_t bits = 0;
    THX_HIP(hipMemcpy(&bits, t->devError, 4, hipMemcpyDeviceToHost));
    if (bits == 0) return THX_OK;
    THX_HIP(hipMemsetAsync(t->devError, 0, 4, t->ctx->stream));
    THX_HIP(hipStreamSynchronize(t->ctx->stream));
    std::string msg = "device error:";
    if (bits & THX_DEVERR_WATCHDOG) {
        msg += " the staged kernel's scheduler watchdog tripped (results of this batch are incomplete";
        const uint32_t why = (bits >> 8) & 31u; // dev_staged.hpp: W_*
        if (why & 1u) msg += "; a wavefront found no work for 2^20 polls while photons were still in flight";
        if (why & 2u) msg += "; an appended ring cell never became free";
        if (why & 4u) msg += "; a claimed ring cell never received its entry";
        if (why & 8u) msg += "; a ring position reached 2^25";
        if (why & 16u) msg += "; THX_TEST_ABORT";
        msg += ")";
    }
    if (bits & ~(THX_DEVERR_WATCHDOG | (31u << 8))) msg += " unknown bits " + std::to_string(bits);
    return fail(THX_ERR_DEVICE, msg);
}

// (second line of defence) a workgroup whose scheduler watchdog tripped also poisons its slab with NaN
static int poisonCheck(const thx_tracer* t, const float* curve, uint32_t n_bins) {
    if (t->staged)
        for (uint32_t i = 0; i < n_bins; ++i)
            if (std::isnan(curve[i])) return fail(THX_ERR_HIP, "the staged kernel's scheduler watchdog tripped: the light curve is invalid");
    return THX_OK;
}

extern "C" {

const char* thx_last_error(void) { return g_error.c_str(); }

int thx_device_count(int* out) {
    if (!out) return fail(THX_ERR_INVALID, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out = 0;
        return fail(THX_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *out = n;
    return THX_OK;
}

int thx_ctx_create(int device, thx_ctx** out) {
    if (!out) return fail(THX_ERR_INVALID, "null argument");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(THX_ERR_NO_DEVICE, "no HIP device available (libtheia_hip has no CPU fallback)");
    if (device < 0 || device >= n) return fail(THX_ERR_INVALID, "device index out of range");
    THX_HIP(hipSetDevice(device));
    thx_ctx* c = new (std::nothrow) thx_ctx();
    if (!c) return fail(THX_ERR_INVALID, "out of host memory");
    c->device = device;
    e = hipGetDeviceProperties(&c->prop, device);
    if (e != hipSuccess) { delete c; return hipFail(e, "hipGetDeviceProperties"); }
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hipFail(e, "hipStreamCreate"); }
    c->lanes[0] = c->stream;
    *out = c;
    return THX_OK;
}

int thx_ctx_destroy(thx_ctx* ctx) {
    if (!ctx) return THX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (uint32_t l = 0; l < THX_MAX_SLOTS; ++l)
        if (ctx->lanes[l]) (void)hipStreamDestroy(ctx->lanes[l]);
    if (ctx->copyStream) (void)hipStreamDestroy(ctx->copyStream);
    delete ctx;
    return THX_OK;
}

int thx_ctx_device_name(thx_ctx* ctx, char* buf, size_t n) {
    if (!ctx || !buf || n == 0) return fail(THX_ERR_INVALID, "null argument");
    std::snprintf(buf, n, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    return THX_OK;
}

int thx_ctx_device_info(thx_ctx* ctx, uint32_t* n_cus, uint32_t* lds_per_cu, uint64_t* hbm_bytes) {
    if (!ctx) return fail(THX_ERR_INVALID, "null argument");
    if (n_cus) *n_cus = (uint32_t)ctx->prop.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (uint32_t)ctx->prop.maxSharedMemoryPerMultiProcessor;
    if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->prop.totalGlobalMem;
    return THX_OK;
}

int thx_ctx_sync(thx_ctx* ctx) {
    if (!ctx) return fail(THX_ERR_INVALID, "null argument");
    THX_HIP(hipSetDevice(ctx->device));
    for (uint32_t l = 0; l < ctx->nLanes; ++l)
        if (ctx->lanes[l]) THX_HIP(hipStreamSynchronize(ctx->lanes[l]));
    return THX_OK;
}

int thx_ctx_concurrent_kernels(thx_ctx* ctx, uint32_t* out) {
    if (!ctx || !out) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(ctx->lock);
    if (ctx->concurrentKernels == 0) {
        THX_HIP(hipSetDevice(ctx->device));
        // eight idle wavefronts of 200 us each, one per lane stream, each noting its start and end on the device's own clock:
        // how many of them were running at the same moment?
        constexpr uint32_t kProbe = THX_MAX_SLOTS;
        constexpr unsigned long long kTicks = 20000; // x 10 ns
        hipStream_t st[kProbe];
        for (uint32_t l = 0; l < kProbe; ++l) THX_HIP(laneStream(ctx, l, &st[l]));
        unsigned long long* d_when = nullptr;
        THX_HIP(hipMalloc((void**)&d_when, sizeof(unsigned long long) * 2 * kProbe));
        hipEvent_t ev[kProbe] = {};
        for (uint32_t l = 0; l < kProbe; ++l) THX_HIP(hipEventCreateWithFlags(&ev[l], hipEventDisableTiming));
        hipError_t e = hipDeviceSynchronize();
        for (uint32_t l = 0; l < kProbe && e == hipSuccess; ++l) e = launchSpin(100, d_when + 2 * l, st[l]); // code object loaded, queues created
        if (e == hipSuccess) e = hipDeviceSynchronize();
        // The question the scheduler asks is "how many LANES can I use": the largest L (16, 12, 8, 6, 4, 3, 2) whose first L lane streams all
        // run at the same moment -- streams that share a hardware queue run one after the other, and which streams share depends on
        // the order in which the process created its streams.
        uint32_t best = 1;
        for (uint32_t lanes : {16u, 12u, 8u, 6u, 4u, 3u, 2u}) {
            if (e != hipSuccess || best > 1) break;
            bool all = true;
            for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {
                // (every stream gets what a batch puts on a lane: a long kernel, a short one behind it, an event)
                for (uint32_t l = 0; l < lanes && e == hipSuccess; ++l) {
                    e = launchSpin(kTicks, d_when + 2 * l, st[l]);
                    if (e == hipSuccess) e = launchSpin(1, nullptr, st[l]);
                    if (e == hipSuccess) e = hipEventRecord(ev[l], st[l]);
                }
                if (e == hipSuccess) e = hipDeviceSynchronize();
                unsigned long long when[2 * kProbe];
                if (e == hipSuccess) e = hipMemcpy(when, d_when, sizeof(when), hipMemcpyDeviceToHost);
                if (e != hipSuccess) break;
                // all of them running when the last one starts?
                unsigned long long lastStart = 0, firstEnd = ~0ull;
                for (uint32_t l = 0; l < lanes; ++l) { lastStart = std::max(lastStart, when[2 * l]); firstEnd = std::min(firstEnd, when[2 * l + 1]); }
                all = all && lastStart < firstEnd;
            }
            if (e == hipSuccess && all) best = lanes;
        }
        (void)hipFree(d_when);
        for (uint32_t l = 0; l < kProbe; ++l) (void)hipEventDestroy(ev[l]);
        if (e != hipSuccess) return hipFail(e, "thx_ctx_concurrent_kernels");
        ctx->concurrentKernels = std::max(1u, std::min(kProbe, best));
        if (std::getenv("THX_VERBOSE")) std::fprintf(stderr, "[thx] lane streams that run side by side: %u\n", ctx->concurrentKernels);
    }
    *out = ctx->concurrentKernels;
    return THX_OK;
}

int thx_mem_alloc(thx_ctx* ctx, size_t nbytes, uint64_t* dev_addr) {
    if (!ctx || !dev_addr) return fail(THX_ERR_INVALID, "null argument");
    THX_HIP(hipSetDevice(ctx->device));
    void* p = nullptr;
    THX_HIP(hipMalloc(&p, nbytes ? nbytes : 16));
    *dev_addr = (uint64_t)(uintptr_t)p;
    return THX_OK;
}

int thx_mem_free(thx_ctx* ctx, uint64_t dev_addr) {
    if (!ctx) return fail(THX_ERR_INVALID, "null argument");
    if (dev_addr == 0) return THX_OK;
    THX_HIP(hipSetDevice(ctx->device));
    THX_HIP(hipFree((void*)(uintptr_t)dev_addr));
    return THX_OK;
}

int thx_host_register(thx_ctx* ctx, void* ptr, size_t nbytes) {
    if (!ctx || !ptr || nbytes == 0) return fail(THX_ERR_INVALID, "null argument");
    THX_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipHostRegister(ptr, nbytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return THX_OK; }
    if (e != hipSuccess) return hipFail(e, "hipHostRegister");
    return THX_OK;
}

int thx_host_unregister(thx_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return fail(THX_ERR_INVALID, "null argument");
    THX_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipHostUnregister(ptr);
    if (e == hipErrorHostMemoryNotRegistered) { (void)hipGetLastError(); return THX_OK; }
    if (e != hipSuccess) return hipFail(e, "hipHostUnregister");
    return THX_OK;
}

int thx_mem_upload(thx_ctx* ctx, uint64_t dev_addr, const void* src, size_t nbytes) {
    if (!ctx || (!src && nbytes) || (dev_addr == 0 && nbytes)) return fail(THX_ERR_INVALID, "null argument");
    if (nbytes == 0) return THX_OK;
    THX_HIP(hipSetDevice(ctx->device));
    // on the context's stream (it is non-blocking: a null-stream copy would not be ordered with
    // launches in flight that still read the old contents) and complete on return (`src` is the caller's)
    if (ctx->nLanes > 1) THX_HIP(hipDeviceSynchronize()); // ... nor with launches on the other lanes
    THX_HIP(hipMemcpyAsync((void*)(uintptr_t)dev_addr, src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    THX_HIP(hipStreamSynchronize(ctx->stream));
    return THX_OK;
}

int thx_mem_download(thx_ctx* ctx, void* dst, uint64_t dev_addr, size_t nbytes) {
    if (!ctx || (!dst && nbytes) || (dev_addr == 0 && nbytes)) return fail(THX_ERR_INVALID, "null argument");
    if (nbytes == 0) return THX_OK;
    THX_HIP(hipSetDevice(ctx->device));
    THX_HIP(hipStreamSynchronize(ctx->stream));
    THX_HIP(hipMemcpy(dst, (const void*)(uintptr_t)dev_addr, nbytes, hipMemcpyDeviceToHost));
    return THX_OK;
}

// ------------------------------------------------------------------- scene

static int uploadVec(thx_scene* s, const void* src, size_t nbytes, void** out) {
    void* p = nullptr;
    THX_HIP(hipMalloc(&p, nbytes ? nbytes : 16));
    s->allocs.push_back(p);
    if (nbytes) THX_HIP(hipMemcpy(p, src, nbytes, hipMemcpyHostToDevice));
    *out = p;
    return THX_OK;
}

int thx_scene_create(thx_ctx* ctx, const thx_mesh_desc* meshes, uint32_t n_meshes,
                     const thx_instance_desc* instances, uint32_t n_instances, thx_scene** out) {
    if (!ctx || !meshes || !instances || !out) return fail(THX_ERR_INVALID, "null argument");
    *out = nullptr;
    THX_HIP(hipSetDevice(ctx->device));
    BuiltScene built;
    try {
        built = buildScene(meshes, n_meshes, instances, n_instances, kTreeDepth);
    } catch (const std::exception& e) {
        return fail(THX_ERR_INVALID, e.what());
    }
    thx_scene* s = new (std::nothrow) thx_scene();
    if (!s) return fail(THX_ERR_INVALID, "out of host memory");
    s->ctx = ctx;
    // one record of padding behind the last triangle: the walk asks for a leaf's triangles two at a time, whether or not
    // the second one belongs to the leaf (dev_walk.hpp: Walker::advance)
    const size_t nTriangles = built.tris.size();
    built.tris.emplace_back();
    std::memset(&built.tris.back(), 0, sizeof(TriRecord));
    int rc;
    void* p;
#define UP(vec, field, T)                                                          \
    rc = uploadVec(s, (vec).data(), (vec).size() * sizeof((vec)[0]), &p);          \
    if (rc != THX_OK) { thx_scene_destroy(s); return rc; }                         \
    s->view.field = reinterpret_cast<const T*>(p);
    UP(built.pairs, nodes, NodePair)
    UP(built.frames, frames, TreeFrame)
    UP(built.tris, tris, TriRecord)
    UP(built.normals, normals, NormalRecord)
    UP(built.instances, instances, InstanceRecord)
    UP(built.walkInstances, walkInstances, WalkInstance)
#undef UP
    // reference-layout vertex / index buffers (scene.types.glsl:6-14), one pair per mesh
    std::vector<uint64_t> vaddr(n_meshes), iaddr(n_meshes);
    for (uint32_t m = 0; m < n_meshes; ++m) {
        rc = uploadVec(s, meshes[m].vertices, sizeof(float) * 6 * (size_t)meshes[m].n_vertices, &p);
        if (rc != THX_OK) { thx_scene_destroy(s); return rc; }
        vaddr[m] = (uint64_t)(uintptr_t)p;
        rc = uploadVec(s, meshes[m].indices, sizeof(uint32_t) * 3 * (size_t)meshes[m].n_triangles, &p);
        if (rc != THX_OK) { thx_scene_destroy(s); return rc; }
        iaddr[m] = (uint64_t)(uintptr_t)p;
    }
    // Geometry[] (scene.py:628-645)
    std::vector<GeometryGLSL> geo(n_instances);
    for (uint32_t k = 0; k < n_instances; ++k) {
        geo[k].vertices = vaddr[instances[k].mesh];
        geo[k].indices = iaddr[instances[k].mesh];
        geo[k].material = instances[k].material;
    }
    rc = uploadVec(s, geo.data(), geo.size() * sizeof(GeometryGLSL), &p);
    if (rc != THX_OK) { thx_scene_destroy(s); return rc; }
    s->view.geometries = reinterpret_cast<const GeometryGLSL*>(p);
    s->view.tlas = built.frames[0];
    s->view.nInstances = n_instances;
    s->view.wide = built.wide ? 1u : 0u;
    s->wide = built.wide;

    s->info.n_nodes = (uint32_t)built.nodes.size();
    s->info.n_triangles = (uint32_t)nTriangles;
    s->info.n_instances = n_instances;
    s->info.max_depth = built.maxDepth;
    s->info.bytes_nodes = built.pairs.size() * sizeof(NodePair) + built.frames.size() * sizeof(TreeFrame);
    s->nPairs = (uint32_t)built.pairs.size();
    s->info.bytes_triangles = nTriangles * (sizeof(TriRecord) + sizeof(NormalRecord));
    s->info.bytes_instances = built.instances.size() * sizeof(InstanceRecord);
    uint64_t total = s->info.bytes_nodes + s->info.bytes_triangles + s->info.bytes_instances +
                     built.walkInstances.size() * sizeof(WalkInstance) + geo.size() * sizeof(GeometryGLSL);
    for (uint32_t m = 0; m < n_meshes; ++m)
        total += 24ull * meshes[m].n_vertices + 12ull * meshes[m].n_triangles;
    s->info.bytes_total = total;
    // what the walk and the hit reconstruction load per counted unit (thx_layout.h; dev_walk.hpp: Walker,
    // processRayQuery): half a NodePair per child box; the three 12-B vertex loads of a triangle test (its TriRecord is 64 B: every
    // vertex is stored as x y z x y and read from byte 0 / 4 / 8); a WalkInstance (80) per instance entry; per reconstructed hit the
    // triangle's three vertices (36) + NormalRecord (48) + Geometry (24) + Material (24) + the instance record (112)
    constexpr uint32_t kTriangleLoadBytes = 36;
    s->info.layout_box_bytes = sizeof(NodePair) / 2;
    s->info.layout_triangle_bytes = kTriangleLoadBytes;
    s->info.layout_instance_bytes = sizeof(WalkInstance);
    s->info.layout_hit_bytes = kTriangleLoadBytes + sizeof(NormalRecord) + sizeof(GeometryGLSL) + sizeof(MaterialGLSL) + sizeof(InstanceRecord);
    *out = s;
    return THX_OK;
}

int thx_scene_destroy(thx_scene* scene) {
    if (!scene) return THX_OK;
    (void)hipSetDevice(scene->ctx->device);
    (void)hipStreamSynchronize(scene->ctx->stream);
    for (void* p : scene->allocs) (void)hipFree(p);
    delete scene;
    return THX_OK;
}

int thx_scene_get_info(const thx_scene* scene, thx_scene_info* out) {
    if (!scene || !out) return fail(THX_ERR_INVALID, "null argument");
    *out = scene->info;
    return THX_OK;
}

int thx_scene_query(thx_scene* scene, const float* rays7, uint32_t n, float* out6) {
    if (!scene || !rays7 || !out6) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(scene->ctx->lock);
    if (n == 0) return THX_OK;
    thx_ctx* ctx = scene->ctx;
    THX_HIP(hipSetDevice(ctx->device));
    float *d_in = nullptr, *d_out = nullptr;
    THX_HIP(hipMalloc((void**)&d_in, sizeof(float) * 7 * (size_t)n));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(float) * 6 * (size_t)n);
    if (e != hipSuccess) { (void)hipFree(d_in); return hipFail(e, "hipMalloc"); }
    e = hipMemcpy(d_in, rays7, sizeof(float) * 7 * (size_t)n, hipMemcpyHostToDevice);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&ev0);
    if (e == hipSuccess) e = hipEventCreate(&ev1);
    if (e == hipSuccess) e = launchQuery(scene->view, d_in, n, d_out, ctx->stream); // warm caches
    if (e == hipSuccess) e = hipEventRecord(ev0, ctx->stream);
    if (e == hipSuccess) e = launchQuery(scene->view, d_in, n, d_out, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(ev1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipEventElapsedTime(&scene->lastQueryMs, ev0, ev1);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (e == hipSuccess) e = hipMemcpy(out6, d_out, sizeof(float) * 6 * (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return hipFail(e, "thx_scene_query");
    return THX_OK;
}

int thx_scene_query_ms(const thx_scene* scene, float* ms) {
    if (!scene || !ms) return fail(THX_ERR_INVALID, "null argument");
    *ms = scene->lastQueryMs;
    return THX_OK;
}

// ------------------------------------------------------------------ tracer

int thx_tracer_create(thx_ctx* ctx, const thx_tracer_desc* desc, thx_scene* scene, thx_tracer** out) {
    if (!ctx || !desc || !out) return fail(THX_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->kind > THX_TRACER_LIGHT_SAMPLER) return fail(THX_ERR_UNSUPPORTED, "unknown tracer kind");
    const bool sampler = desc->kind == THX_TRACER_LIGHT_SAMPLER;
    const bool replay = desc->kind == THX_TRACER_HIT_REPLAY || sampler; // no tracing kernel
    if (((desc->flags & THX_FLAG_LIGHT_HOST) != 0) != (desc->light_queue_capacity != 0) ||
        ((desc->flags & THX_FLAG_WAVELENGTH_HOST) != 0) != (desc->wavelength_queue_capacity != 0))
        return fail(THX_ERR_INVALID, "host light / wavelength source flags and queue capacities do not match");
    if (sampler && (((desc->flags & THX_FLAG_LIGHT_HOST) && ((desc->flags & THX_FLAG_LIGHT_QUEUE_POLARIZED) != 0) != ((desc->flags & THX_FLAG_POLARIZED) != 0))))
        return fail(THX_ERR_INVALID, "mismatch in light queue definition (lightsource.sample.glsl:28-33)");
    if (desc->flags & THX_FLAG_SOURCES_PRESAMPLED) return fail(THX_ERR_INVALID, "THX_FLAG_SOURCES_PRESAMPLED is set by the library, not by the caller");
    if ((desc->flags & THX_FLAG_LIGHT_KIND_MASK) > THX_FLAG_LIGHT_CASCADE) return fail(THX_ERR_UNSUPPORTED, "unknown light source kind");
    if ((desc->flags & THX_FLAG_LIGHT_KIND_MASK) && (desc->flags & (THX_FLAG_LIGHT_HOST | THX_FLAG_LIGHT_CONE | THX_FLAG_LIGHT_PENCIL)))
        return fail(THX_ERR_INVALID, "a tracer has one light source: a Cherenkov-family kind excludes the host / cone / pencil flags");
    if (desc->kind == THX_TRACER_SCENE_FORWARD && !scene)
        return fail(THX_ERR_INVALID, "scene tracer needs a scene");
    const bool record = !sampler && (desc->response_kind == THX_RESPONSE_RECORD || desc->response_kind == THX_RESPONSE_STORE_TIME);
    if (desc->response_kind > THX_RESPONSE_STORE_TIME) return fail(THX_ERR_INVALID, "unknown response kind");
    if (record && desc->hit_capacity == 0) return fail(THX_ERR_INVALID, "cannot create a hit queue of size 0 (response.py:266-267)");
    if (!record && (desc->n_bins == 0 || desc->n_bins > 8192))
        return fail(THX_ERR_INVALID, "n_bins must be in [1, 8192] (LDS histogram, response.py:1098)");
    if (desc->capacity == 0) return fail(THX_ERR_INVALID, "capacity must be positive");
    THX_HIP(hipSetDevice(ctx->device));
    thx_tracer* t = new (std::nothrow) thx_tracer();
    if (!t) return fail(THX_ERR_INVALID, "out of host memory");
    t->ctx = ctx;
    t->scene = scene;
    t->desc = *desc;
    if (record || sampler) t->desc.n_bins = 1; // the kernels keep their (unused) one-bin histogram
    if (sampler) t->desc.hit_capacity = desc->capacity;
    desc = &t->desc;
    const bool pol = (desc->flags & THX_FLAG_POLARIZED) != 0;
    const bool mis = (desc->flags & THX_FLAG_DISABLE_MIS) == 0;
    if (!replay) t->variant = traceVariant(desc->kind, pol, mis);
    // the record response and the Sobol sampler are compiled into the instrumented variants only
    t->debug = record || (desc->flags & (THX_FLAG_STATISTICS | THX_FLAG_TRACK | THX_FLAG_RNG_SOBOL)) != 0;
    t->block = replay ? kReplayBlock : kTraceBlock;
    if (scene) {
        t->stackDepth = std::min<uint32_t>(kStackDepth, scene->info.max_depth + 2);
        // 16-bit stack entries are read sign-extended (0xFFFE / 0xFFFF = leave the instance / done): every ref = index << 3 | count
        // of the scene has to stay below 0x8000
        const uint32_t recordSlots = scene->wide ? 2u : kRecordSlots;
        t->stack16 = !scene->wide && scene->nPairs < 4096 * recordSlots && scene->info.n_triangles < 4096 && scene->info.n_instances < 4096;
        // (what the bound stands for, checked on the value itself: the largest ref any stack entry of this scene can hold)
        const uint32_t maxIndex = std::max(std::max(scene->nPairs / recordSlots, scene->info.n_triangles), scene->info.n_instances);
        if (t->stack16 && ((maxIndex << 3) | 7u) >= 0x8000u) { delete t; return fail(THX_ERR_INVALID, "internal: 16-bit stack entries picked for a scene with refs >= 0x8000"); }
        if (const char* env = std::getenv("THX_STACK32")) t->stack16 = t->stack16 && std::atoi(env) == 0;
        // test hook: a stack shallower than the scene needs (the walk then flags ERROR_CODE_TRACE_ABORT)
        if (const char* env = std::getenv("THX_STACK_DEPTH")) t->stackDepth = (uint32_t)std::min<int>(kStackDepth, std::max(1, std::atoi(env)));
    }
    // kernel form: the staged scene kernel unless THX_KERNEL=lane asks for the lane kernel
    // (kept as the simple cross-check of the staged one) or there is no step to stage
    t->staged = desc->kind == THX_TRACER_SCENE_FORWARD && scene && desc->path_length >= 1;
    if (const char* env = std::getenv("THX_KERNEL")) {
        if (std::strcmp(env, "lane") == 0) t->staged = false;
    }
    t->mode = (t->debug ? 1u : 0u) | (t->staged ? 2u : 0u);
    if (t->staged) {
        const uint32_t fullBlock = t->debug ? (scene->wide && pol ? kStagedBlockDebugWidePol : kStagedBlockDebug) : kStagedBlock;
        t->block = fullBlock;
        if (const char* env = std::getenv("THX_THREADS")) t->block = (uint32_t)std::min<int>((int)fullBlock, std::max(64, std::atoi(env) / 64 * 64));
        size_t budget = (size_t)ctx->prop.maxSharedMemoryPerMultiProcessor - 2048; // whole CU, one workgroup
        if (const char* env = std::getenv("THX_POOL_LDS_KB")) budget = (size_t)std::max(16, std::atoi(env)) * 1024;
        // Deep hierarchies: rays use the worst-case depth rarely, so only the first kStackLds entries (THX_STACK_LDS)
        // come out of LDS and the rest lives in global memory (dev_walk.hpp: Walker::push)
        t->stackLds = std::min(t->stackDepth, t->stack16 ? kStackLds16 : (scene->wide ? kStackLds32Wide : kStackLds32));
        if (const char* env = std::getenv("THX_STACK_LDS")) t->stackLds = (uint32_t)std::min<int>((int)t->stackDepth, std::max(1, std::atoi(env)));
        // one more slot per lane below entry 0 (dev_walk.hpp: Walker<V, FAST>)
        // (the variants compiled for a stack that is all in LDS do not check for a full stack: only where it cannot fill up)
        const bool stackAllLds = t->stack16 && t->stackLds == t->stackDepth && t->stackDepth >= scene->info.max_depth + 2;
        t->stackSlots = t->stackLds + 1u + (scene->wide ? 1u : 0u); // (4-wide walks: one more slot, beyond the LDS part, for the writes of visitStraight4 that nobody reads)
        const uint32_t stackBytes = t->stackSlots * (t->stack16 ? 2u : 4u);
        // The pool comes first: measured on MI355X, 768 -> 1024 photons per CU is worth 14 %.
        // Deep scenes (large per-thread stacks) leave a pool much smaller than the workgroup:
        // batches run half empty then. Fewer wavefronts with fuller batches win while the pool
        // holds less than 3/4 photon per thread.
        // Rings of exactly the pool's size (rounded up to 64) instead of the next power of two: their positions are turned
        // into cells by a reciprocal multiplication that is exact below 2^25 appends per ring and launch; a generous bound
        // on those -- photons of the busiest workgroup x steps x 16 (two shadow rays, walks set aside and taken up again)
        // -- decides (THX_RING_POW2=1 forces the power of two). The kernel checks its ring counters against the limit.
        // (decided below, once pool and grid are final)
        // Bound from the values a launch actually uses: a workgroup traces ceil(batch / grid) photons with grid = min(CUs, ceil(batch
        // / pool)), i.e. at most max(ceil(capacity / CUs), the largest pool) -- kPoolMax covers THX_POOL and every pool the
        // sizing below can pick.
        {
            const uint64_t cus = (uint64_t)std::max(1, ctx->prop.multiProcessorCount);
            const uint64_t perWg = std::max<uint64_t>(((uint64_t)desc->capacity + cus - 1) / cus, kPoolMax);
            t->exactRings = perWg * (uint64_t)(desc->path_length + 2) * 16u < (1ull << 25);
            if (const char* env = std::getenv("THX_RING_POW2")) t->exactRings = t->exactRings && std::atoi(env) == 0;
        }
        const bool xr = t->exactRings;
        t->pool = stagedPoolFor(pol, desc->n_bins, stackBytes, budget, t->block, 0, xr);
        if (!std::getenv("THX_THREADS"))
            while (t->block > 512 && 4u * t->pool < 3u * t->block) {
                t->block -= 128;
                t->pool = stagedPoolFor(pol, desc->n_bins, stackBytes, budget, t->block, 0, xr);
            }
        t->pool = std::min(t->pool, kPoolBest);
        if (scene->wide) t->mode |= 32u; // the variant that walks 4-wide records (32-bit entries)
        if (t->stack16) t->mode |= 4u; // the kernel variant with 16-bit stack entries compiled in ...
        if (stackAllLds) t->mode |= 8u; // ... and with the whole stack in LDS
        // Small scenes: the walk reads its pair records from a copy in LDS if that leaves a pool of kLdsNodeMinPool photons
        // (THX_LDS_NODES=0 / 1: never / whenever it fits at all)
        if ((t->mode & 4u) && t->block == fullBlock && !scene->wide) { // (16-bit stack entries: the variants that exist with the copy)
            const size_t nodeBytes = (size_t)scene->nPairs * sizeof(NodePair);
            const uint32_t poolN = std::min(stagedPoolFor(pol, desc->n_bins, stackBytes, budget, t->block, nodeBytes, xr), kPoolBest);
            bool use = nodeBytes <= kLdsNodeMaxBytes && poolN >= kLdsNodeMinPool;
            if (const char* env = std::getenv("THX_LDS_NODES")) use = std::atoi(env) != 0 && nodeBytes + 32 * 1024 <= budget && poolN >= 128;
            if (use) {
                t->mode |= 16u;
                t->pool = poolN;
                t->ldsNodeBytes = nodeBytes;
            }
        }
        if (const char* env = std::getenv("THX_POOL")) t->pool = (uint32_t)std::min(2048, std::max(64, std::atoi(env)));
        t->ringCap = stagedRingCap(t->pool, xr);
        t->ldsNodeOffset = stagedLdsBytes(pol, desc->n_bins, stackBytes, t->pool, t->block, 0, xr);
        t->lds = stagedLdsBytes(pol, desc->n_bins, stackBytes, t->pool, t->block, t->ldsNodeBytes, xr);
        if (t->lds > budget + 2048) { delete t; return fail(THX_ERR_INVALID, "THX_POOL does not fit the LDS of a compute unit"); }
    } else {
        t->stackLds = t->stackDepth;
        t->lds = traceLdsBytes(desc->kind, desc->n_bins, t->block, t->stackDepth, t->stack16);
    }
    if (std::getenv("THX_VERBOSE"))
        std::fprintf(stderr, "[thx] tracer kind=%u pol=%d staged=%d block=%u lds=%zu pool=%u rings=%u stackDepth=%u (LDS %u) stack16=%d ldsNodes=%zu B\n",
                     (unsigned)desc->kind, (int)pol, (int)t->staged, t->block, t->lds, t->pool, t->ringCap, t->stackDepth, t->stackLds, (int)t->stack16, t->ldsNodeBytes);
    // replay streams its queue once: a few workgroups per CU hide the load latency
    int perCU = replay ? 8 : t->variant.occupancy(t->mode, t->block, t->lds);
    if (perCU <= 0) { delete t; return fail(THX_ERR_HIP, "kernel does not fit on the device (occupancy 0)"); }
    uint32_t mult = 1;
    if (const char* env = std::getenv("THX_GRID_MULT")) mult = (uint32_t)std::max(1, std::atoi(env));
    t->gridMax = (uint32_t)ctx->prop.multiProcessorCount * (uint32_t)perCU * mult;
    if (const char* env = std::getenv("THX_REFILL")) t->refillBelow = (uint32_t)std::min(63, std::max(0, std::atoi(env)));
    for (uint32_t i = 0; i < 2; ++i)
        if (int rc = ensureSlot(t, i)) { thx_tracer_destroy(t); return rc; }
    hipError_t e = hipSuccess;
    for (uint32_t i = 0; i < thx_tracer::kRing && e == hipSuccess; ++i) {
        e = hipEventCreate(&t->evStart[i]);
        if (e == hipSuccess) e = hipEventCreate(&t->evStop[i]);
    }
    if (e == hipSuccess && t->staged) {
        // scratch of the workgroups (words per workgroup; lane l of `conc` launches side by side uses the workgroups from l * gridMax / conc on)
        t->coldWg = stagedColdBytes(pol, t->pool) / 4;
        t->spillWg = 64u * (t->block / 64u) * (size_t)(t->stackDepth - t->stackLds);
        t->parkWg = 2u * (8u + std::max(t->stackDepth, 8u)) * (size_t)t->pool; // (a walk is taken up again with eight entries at once)
        e = allocScratch(t, t->gridMax);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&t->stats, sizeof(unsigned long long) * THX_N_STATS);
    if (e == hipSuccess) e = hipMemset(t->stats, 0, sizeof(unsigned long long) * THX_N_STATS);
    if (e == hipSuccess) e = hipMalloc((void**)&t->devError, 4);
    if (e == hipSuccess) e = hipMemset(t->devError, 0, 4);
    if (e == hipSuccess) e = hipMalloc((void**)&t->prof, sizeof(unsigned long long) * 48);
    if (e == hipSuccess) e = hipMemset(t->prof, 0, sizeof(unsigned long long) * 48);
    if (e == hipSuccess && (desc->flags & THX_FLAG_TRACK)) {
        t->trackEvents = desc->path_length + 3;
        size_t nb = sizeof(float) * THX_TRACK_WORDS * (size_t)t->trackEvents * desc->capacity;
        e = hipMalloc((void**)&t->track, nb);
        if (e == hipSuccess) e = hipMemset(t->track, 0, nb);
    }
    if (e == hipSuccess && record) {
        const size_t nb = 4 + hitColumns(*desc) * 4 * (size_t)desc->hit_capacity;
        e = hipMalloc((void**)&t->hits, nb);
        if (e == hipSuccess) e = hipMemset(t->hits, 0, nb);
        if (e == hipSuccess && !replay) { // one queue per pipeline config
            e = hipMalloc((void**)&t->hits1, nb);
            if (e == hipSuccess) e = hipMemset(t->hits1, 0, nb);
            for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&t->evHits[i], hipEventDisableTiming);
        }
    }
    if (e == hipSuccess && sampler) {
        const size_t nb = 4 + hitColumns(*desc) * 4 * (size_t)desc->capacity, nl = 4 + 2 * 4 * (size_t)desc->capacity;
        e = hipMalloc((void**)&t->hits, nb);
        if (e == hipSuccess) e = hipMemset(t->hits, 0, nb);
        if (e == hipSuccess) e = hipMalloc((void**)&t->lamOut, nl);
        if (e == hipSuccess) e = hipMemset(t->lamOut, 0, nl);
    }
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        if (desc->light_queue_capacity) {
            const size_t nb = ((desc->flags & THX_FLAG_LIGHT_QUEUE_POLARIZED) ? 15 : 8) * 4 * (size_t)desc->light_queue_capacity;
            e = hipMalloc((void**)&t->lightIn[i], nb);
            if (e == hipSuccess) e = hipMemset(t->lightIn[i], 0, nb);
        }
        if (e == hipSuccess && desc->wavelength_queue_capacity) {
            const size_t nb = 2 * 4 * (size_t)desc->wavelength_queue_capacity;
            e = hipMalloc((void**)&t->lamIn[i], nb);
            if (e == hipSuccess) e = hipMemset(t->lamIn[i], 0, nb);
        }
    }
    if (e == hipSuccess && desc->kind == THX_TRACER_HIT_REPLAY) {
        const size_t nb = thx_hit_queue_bytes(desc->capacity, pol);
        e = hipMalloc((void**)&t->replayIn, nb);
        if (e == hipSuccess) e = hipMemset(t->replayIn, 0, nb);
    }
    // the memsets above run on the null stream, which is NOT ordered with the context's
    // non-blocking stream: finish them before anything can be launched
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { thx_tracer_destroy(t); return hipFail(e, "thx_tracer_create"); }
    *out = t;
    return THX_OK;
}

int thx_tracer_destroy(thx_tracer* t) {
    if (!t) return THX_OK;
    (void)hipSetDevice(t->ctx->device);
    (void)hipDeviceSynchronize();
    for (uint32_t i = 0; i < THX_MAX_SLOTS; ++i) {
        if (t->slabs[i]) (void)hipFree(t->slabs[i]);
        if (t->out[i]) (void)hipFree(t->out[i]);
        if (t->pinned[i]) (void)hipHostFree(t->pinned[i]);
        if (t->evFetch[i]) (void)hipEventDestroy(t->evFetch[i]);
    }
    for (uint32_t i = 0; i < thx_tracer::kRing; ++i) {
        if (t->evStart[i]) (void)hipEventDestroy(t->evStart[i]);
        if (t->evStop[i]) (void)hipEventDestroy(t->evStop[i]);
    }
    if (t->stats) (void)hipFree(t->stats);
    if (t->cold) (void)hipFree(t->cold);
    if (t->spill) (void)hipFree(t->spill);
    if (t->park) (void)hipFree(t->park);
    if (t->prof) (void)hipFree(t->prof);
    if (t->devError) (void)hipFree(t->devError);
    if (t->track) (void)hipFree(t->track);
    if (t->hits) (void)hipFree(t->hits);
    if (t->hits1) (void)hipFree(t->hits1);
    for (int i = 0; i < 2; ++i)
        if (t->evHits[i]) (void)hipEventDestroy(t->evHits[i]);
    if (t->replayIn) (void)hipFree(t->replayIn);
    if (t->lamOut) (void)hipFree(t->lamOut);
    for (uint32_t l = 0; l < THX_MAX_SLOTS; ++l) {
        if (t->preLight[l]) (void)hipFree(t->preLight[l]);
        if (t->preLam[l]) (void)hipFree(t->preLam[l]);
    }
    for (int i = 0; i < 2; ++i) {
        if (t->lightIn[i]) (void)hipFree(t->lightIn[i]);
        if (t->lamIn[i]) (void)hipFree(t->lamIn[i]);
    }
    delete t;
    return THX_OK;
}

static void launchGeometry(const thx_tracer* t, uint32_t batch, uint32_t* grid, uint32_t* ppw) {
    const uint32_t gridMax = t->laneGrid(); // workgroups one launch may take
    if (t->desc.kind == THX_TRACER_HIT_REPLAY || t->desc.kind == THX_TRACER_LIGHT_SAMPLER) { // one hit per thread until the chip is full, then grid-stride
        const uint64_t need = ((uint64_t)t->desc.capacity + t->block - 1) / t->block;
        *grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(gridMax, need));
        *ppw = 0;
        return;
    }
    if (t->staged) { // one pool-owning workgroup per CU; each gets a contiguous photon range
        uint64_t need = ((uint64_t)batch + t->pool - 1) / t->pool;
        *grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(gridMax, need));
        *ppw = 0;
        return;
    }
    const uint32_t wavesPerBlock = t->block / 64;
    uint64_t wavesNeeded = ((uint64_t)batch + 63) / 64;
    uint64_t blocksNeeded = (wavesNeeded + wavesPerBlock - 1) / wavesPerBlock;
    uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(gridMax, blocksNeeded));
    uint64_t waves = (uint64_t)g * wavesPerBlock;
    uint32_t p = (uint32_t)(((uint64_t)batch + waves - 1) / waves);
    *grid = g;
    *ppw = std::max<uint32_t>(p, 1);
}

int thx_tracer_launch_info(thx_tracer* t, uint32_t batch, uint32_t* grid, uint32_t* block, uint32_t* lds_bytes) {
    if (!t) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    uint32_t g, p;
    launchGeometry(t, batch, &g, &p);
    if (grid) *grid = g;
    if (block) *block = t->block;
    if (lds_bytes) *lds_bytes = (uint32_t)t->lds;
    return THX_OK;
}

int thx_tracer_set_concurrency(thx_tracer* t, uint32_t lanes, uint32_t grid_divisor) {
    if (!t) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (lanes < 1 || lanes > THX_MAX_SLOTS) return fail(THX_ERR_INVALID, "lanes must be in [1, THX_MAX_SLOTS]");
    if (grid_divisor < 1 || grid_divisor > lanes) return fail(THX_ERR_INVALID, "grid_divisor must be in [1, lanes]");
    if (lanes > 1 && (t->hits || t->track || t->replayIn || t->lamOut))
        return fail(THX_ERR_INVALID, "tracers with one output queue or track buffer cannot run batches side by side");
    // rings of exactly the pool's size turn positions into cells exactly below 2^25 appends per ring and launch (thx_tracer_create):
    // the same generous bound, with the photons of a workgroup of a launch on 1 / grid_divisor of the chip
    if (t->staged && t->exactRings) {
        const uint64_t grid = std::max<uint64_t>(1, (t->gridMax + grid_divisor - 1) / grid_divisor);
        const uint64_t perWg = std::max<uint64_t>(((uint64_t)t->desc.capacity + grid - 1) / grid, kPoolMax);
        if (perWg * (uint64_t)(t->desc.path_length + 2) * 16u >= (1ull << 25))
            return fail(THX_ERR_INVALID, "grid_divisor too large for this tracer's capacity: a workgroup's ring positions could pass 2^25 (use a smaller divisor or capacity)");
    }
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize()); // the lanes' shares of the scratch memory change
    const uint32_t oldLanes = t->conc, oldDiv = t->gridDiv;
    t->conc = lanes;
    t->gridDiv = grid_divisor;
    if (t->staged) {
        hipError_t e = allocScratch(t, lanes * t->laneGrid());
        if (e != hipSuccess) {
            t->conc = oldLanes; t->gridDiv = oldDiv;
            (void)allocScratch(t, t->conc * t->laneGrid());
            return hipFail(e, "thx_tracer_set_concurrency: scratch memory");
        }
    }
    return THX_OK;
}

int thx_tracer_run(thx_tracer* t, uint32_t cfg, const thx_run_params* params, void* stream) {
    if (!t || !params) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (cfg >= THX_MAX_SLOTS) return fail(THX_ERR_INVALID, "slot index out of range (THX_MAX_SLOTS)");
    const uint32_t batch = params->dispatch.batchSize;
    if (batch > t->desc.capacity) return fail(THX_ERR_INVALID, "batchSize cannot be larger than capacity");
    thx_ctx* ctx = t->ctx;
    THX_HIP(hipSetDevice(ctx->device));
    if (int rc = ensureSlot(t, cfg)) return rc;
    const uint32_t lane = cfg % t->conc;
    hipStream_t s = (hipStream_t)stream;
    if (!s) THX_HIP(laneStream(ctx, lane, &s));
    const uint32_t hostCfg = cfg & 1u; // device queues of the host-fed sources exist for the two pipeline configs
    if ((t->lightIn[0] || t->lamIn[0]) && cfg > 1) return fail(THX_ERR_INVALID, "tracers fed from host queues run in slots 0 and 1 only");

    LaunchArgs a;
    std::memset(&a, 0, sizeof(a));
    a.p = *params;
    a.flags = t->desc.flags;
    if (std::getenv("THX_TEST_ABORT")) a.flags |= 1u << 31; // test hook: the staged kernel starts with its abort flag set
    a.path_length = t->desc.path_length;
    a.n_bins = t->desc.n_bins;
    a.track_events = t->trackEvents;
    a.stack_depth = t->stackDepth;
    a.stack_lds = t->stackLds;
    a.stack_wave_words = (t->staged ? t->stackSlots : t->stackLds) * (t->stack16 ? 32u : 64u);
    // scratch of this lane's workgroups
    const size_t wg0 = (size_t)lane * t->laneGrid();
    a.spill = t->spill ? t->spill + wg0 * t->spillWg : nullptr;
    a.park = t->park ? t->park + wg0 * t->parkWg : nullptr;
    a.park_below = t->scene && t->scene->info.n_instances > kParkInstancedFrom ? kParkBelowInstanced
                   : (t->stackDepth > kParkDeepFrom ? kParkBelowDeep : kParkBelow);
    a.park_rounds = kParkRounds;
    if (const char* env = std::getenv("THX_PARK_BELOW")) a.park_below = (uint32_t)std::min(65, std::max(0, std::atoi(env)));
    if (const char* env = std::getenv("THX_PARK_ROUNDS")) a.park_rounds = (uint32_t)std::max(1, std::atoi(env));
    a.stack16 = t->stack16 ? 1u : 0u;
    a.lds_node_word = (uint32_t)(t->ldsNodeOffset / 4);
    a.lds_node_pairs = t->ldsNodeBytes ? t->scene->nPairs : 0u;
    a.refill_below = t->refillBelow;
    a.pool = t->pool;
    a.ring_cap = t->ringCap;
    if (t->ringCap) { // ceil(2^37 / d): floor(pos * magic / 2^37) == pos / d for pos < 2^37 / d (>= 2^25: d <= 4096)
        a.ring_magic = (uint32_t)(((1ull << 37) + t->ringCap - 1) / t->ringCap);
        a.ring_magic_ts = (uint32_t)(((1ull << 37) + 2ull * t->ringCap - 1) / (2ull * t->ringCap));
    }
    a.slabs = t->slabs[cfg];
    a.cold = t->cold ? t->cold + wg0 * t->coldWg : nullptr;
    a.stats = t->stats;
    a.error = t->devError;
    a.prof = t->prof;
    if (t->hits1 && cfg > 1) return fail(THX_ERR_INVALID, "tracers with a hit queue run in slots 0 and 1 only (one queue per pipeline config)");
    a.hits = (t->hits1 && cfg == 1) ? t->hits1 : t->hits;
    a.hit_capacity = t->desc.hit_capacity;
    a.hit_polarized = t->desc.hit_polarized;
    a.hit_kind = t->desc.response_kind;
    a.light_queue = t->lightIn[hostCfg];
    a.wavelength_queue = t->lamIn[hostCfg];
    a.light_capacity = t->desc.light_queue_capacity;
    a.wavelength_capacity = t->desc.wavelength_queue_capacity;
    const bool tracing = t->desc.kind == THX_TRACER_VOLUME_FORWARD || t->desc.kind == THX_TRACER_SCENE_FORWARD;
    const bool presample = tracing && (t->desc.flags & THX_FLAG_LIGHT_KIND_MASK) != 0u;
    // (one pair of queues per lane: a lane's batches run one after the other on its stream. A caller that brings its own stream
    // orders the batches itself: a pair per slot then)
    const uint32_t pq = stream ? cfg : lane;
    if (presample && !t->preLight[pq]) { // first batch of this lane
        const bool pol = (t->desc.flags & THX_FLAG_POLARIZED) != 0u;
        THX_HIP(hipMalloc((void**)&t->preLight[pq], 4 + (pol ? 15 : 8) * 4 * (size_t)t->desc.capacity));
        THX_HIP(hipMalloc((void**)&t->preLam[pq], 4 + 3 * 4 * (size_t)t->desc.capacity));
    }
    if ((t->lightIn[hostCfg] && batch > t->desc.light_queue_capacity) || (t->lamIn[hostCfg] && batch > t->desc.wavelength_queue_capacity))
        return fail(THX_ERR_INVALID, "batchSize exceeds the capacity of the host light / wavelength source");
    a.track = t->track;
    if (t->scene) a.scene = t->scene->view;
    uint32_t grid, ppw;
    launchGeometry(t, batch, &grid, &ppw);
    a.photons_per_wave = ppw;
    t->lastGrid[cfg] = grid;
    t->lastBatch = batch;
    // (launches side by side are not timed: an event pair per launch costs two more packets per batch on every lane, and the
    // durations of overlapping launches say little)
    const bool timed = t->conc == 1;
    const uint32_t slot = (uint32_t)(t->launches % thx_tracer::kRing);
    if (timed) THX_HIP(hipEventRecord(t->evStart[slot], s));
    if (t->desc.kind == THX_TRACER_LIGHT_SAMPLER) {
        if ((uint64_t)batch + params->photon_offset > UINT32_MAX ||
            (t->lightIn[hostCfg] && batch + params->photon_offset > t->desc.light_queue_capacity) ||
            (t->lamIn[hostCfg] && batch + params->photon_offset > t->desc.wavelength_queue_capacity))
            return fail(THX_ERR_INVALID, "count + baseCount exceeds the capacity of the host light / wavelength source");
        THX_HIP(launchSampler(a, (t->desc.flags & THX_FLAG_POLARIZED) != 0, t->hits, t->lamOut, t->desc.capacity, false, s));
    } else if (t->desc.kind == THX_TRACER_HIT_REPLAY)
        THX_HIP(launchReplay(a, t->replayIn, t->desc.capacity, (t->desc.flags & THX_FLAG_POLARIZED) != 0, grid, s));
    else {
        if (presample) {
            // the sources first, on the same stream: sample idx + photon_offset of wavelength and light source into item idx of the
            // lane's queues (what a LightSampler stage does, lightsource.sample.glsl:52-75), in the photon's start medium ...
            LaunchArgs sa = a;
            const uint32_t* medium = t->desc.kind == THX_TRACER_SCENE_FORWARD ? params->scene.sceneMedium : params->volume.medium;
            sa.p.volume.medium[0] = medium[0];
            sa.p.volume.medium[1] = medium[1];
            const bool pol = (t->desc.flags & THX_FLAG_POLARIZED) != 0u;
            THX_HIP(launchSampler(sa, pol, t->preLight[pq], t->preLam[pq], t->desc.capacity, true, s));
            // ... which the tracing kernel then reads like queues uploaded by the host, RNG dimension included
            a.flags = (a.flags & ~(uint32_t)(THX_FLAG_LIGHT_KIND_MASK | THX_FLAG_FRANK_TAMM_PHOTONS | THX_FLAG_FRANK_TAMM_IS |
                                             THX_FLAG_WAVELENGTH_UNIFORM | THX_FLAG_WAVELENGTH_FUNCTION)) |
                      THX_FLAG_LIGHT_HOST | THX_FLAG_WAVELENGTH_HOST | THX_FLAG_SOURCES_PRESAMPLED | (pol ? THX_FLAG_LIGHT_QUEUE_POLARIZED : 0u);
            a.light_queue = reinterpret_cast<const float*>(t->preLight[pq] + 1);
            a.wavelength_queue = reinterpret_cast<const float*>(t->preLam[pq] + 1);
            a.light_capacity = a.wavelength_capacity = t->desc.capacity;
        }
        THX_HIP(t->variant.launch(t->mode, a, grid, t->block, t->lds, s));
    }
    if (t->hits1) {
        THX_HIP(hipEventRecord(t->evHits[cfg], s));
        t->lastHitCfg = cfg;
    }
    if (timed) {
        THX_HIP(hipEventRecord(t->evStop[slot], s));
        t->lastSlot[cfg] = (int)slot;
        t->launches += 1;
    }
    return THX_OK;
}

int thx_tracer_reduce(thx_tracer* t, uint32_t cfg, float norm, uint64_t d_out, void* stream) {
    if (!t) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (cfg >= THX_MAX_SLOTS || !t->slabs[cfg]) return fail(THX_ERR_INVALID, "no batch was launched into this slot");
    thx_ctx* ctx = t->ctx;
    THX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (!s) THX_HIP(laneStream(ctx, cfg % t->conc, &s));
    float* out = d_out ? (float*)(uintptr_t)d_out : t->out[cfg];
    // into the tracer's own buffer: also into the slot's pinned host buffer, so that thx_tracer_fetch_hist needs no copy command
    THX_HIP(launchReduce(t->slabs[cfg], out, t->desc.n_bins, t->lastGrid[cfg], norm, d_out ? nullptr : t->devError, d_out ? nullptr : t->pinnedDev[cfg], s));
    t->mirrored[cfg] = d_out == 0;
    return THX_OK;
}

int thx_tracer_read_hist(thx_tracer* t, uint32_t cfg, float* out, uint32_t n_bins) {
    if (!t || !out) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (cfg >= THX_MAX_SLOTS || !t->out[cfg] || n_bins != t->desc.n_bins) return fail(THX_ERR_INVALID, "bad slot index or bin count");
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize());
    if (int rc = deviceErrorCheck(t)) return rc;
    THX_HIP(hipMemcpy(out, t->out[cfg], sizeof(float) * n_bins, hipMemcpyDeviceToHost));
    return poisonCheck(t, out, n_bins);
}

int thx_tracer_fetch_hist(thx_tracer* t, uint32_t cfg, void* stream) {
    if (!t) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (cfg >= THX_MAX_SLOTS || !t->out[cfg]) return fail(THX_ERR_INVALID, "no batch was launched into this slot");
    thx_ctx* ctx = t->ctx;
    THX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (!s) THX_HIP(laneStream(ctx, cfg % t->conc, &s));
    // (the reduction wrote curve + device error word into the pinned buffer itself; after a reduction into an external buffer the
    // slot's own curve is copied)
    if (!t->mirrored[cfg]) THX_HIP(hipMemcpyAsync(t->pinned[cfg], t->out[cfg], sizeof(float) * (t->desc.n_bins + 1u), hipMemcpyDeviceToHost, s));
    THX_HIP(hipEventRecord(t->evFetch[cfg], s));
    t->fetched[cfg] = true;
    return THX_OK;
}

int thx_tracer_collect_hist(thx_tracer* t, uint32_t cfg, float* out, uint32_t n_bins) {
    if (!t || !out) return fail(THX_ERR_INVALID, "null argument");
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> guard(t->ctx->lock);
        if (cfg >= THX_MAX_SLOTS || !t->fetched[cfg] || n_bins != t->desc.n_bins)
            return fail(THX_ERR_INVALID, "bad slot index or bin count, or no thx_tracer_fetch_hist to collect");
        ev = t->evFetch[cfg];
    }
    // the wait itself happens outside the context's lock: launches into other slots go on meanwhile
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipEventSynchronize(ev));
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    t->fetched[cfg] = false;
    const size_t nb = sizeof(float) * n_bins;
    uint32_t bits = 0;
    std::memcpy(&bits, (const char*)t->pinned[cfg] + nb, 4);
    if (bits != 0) { // the word is shared by the batches in flight: whoever sees it first reports (and clears) it
        THX_HIP(hipDeviceSynchronize());
        return deviceErrorCheck(t) == THX_OK ? fail(THX_ERR_DEVICE, "device error (already reported by another slot)") : THX_ERR_DEVICE;
    }
    std::memcpy(out, t->pinned[cfg], nb);
    return poisonCheck(t, out, n_bins);
}

int thx_tracer_check(thx_tracer* t) {
    if (!t) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize());
    return deviceErrorCheck(t);
}

int thx_tracer_hist_device_ptr(thx_tracer* t, uint32_t cfg, uint64_t* out) {
    if (!t || !out || cfg >= THX_MAX_SLOTS) return fail(THX_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (int rc = ensureSlot(t, cfg)) return rc;
    *out = (uint64_t)(uintptr_t)t->out[cfg];
    return THX_OK;
}

int thx_tracer_read_stats(thx_tracer* t, uint64_t* out, uint32_t n) {
    if (!t || !out) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (n > THX_N_STATS) n = THX_N_STATS;
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize());
    if (int rc = deviceErrorCheck(t)) return rc;
    THX_HIP(hipMemcpy(out, t->stats, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
    return THX_OK;
}

int thx_tracer_read_stage_profile(thx_tracer* t, uint64_t* out48) {
    if (!t || !out48) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize());
    THX_HIP(hipMemcpy(out48, t->prof, sizeof(uint64_t) * 48, hipMemcpyDeviceToHost));
    THX_HIP(hipMemset(t->prof, 0, sizeof(uint64_t) * 48));
    THX_HIP(hipDeviceSynchronize()); // null-stream memset vs. launches on the non-blocking stream
    return THX_OK;
}

int thx_tracer_reset_stats(thx_tracer* t) {
    if (!t) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize());
    THX_HIP(hipMemset(t->stats, 0, sizeof(unsigned long long) * THX_N_STATS));
    THX_HIP(hipDeviceSynchronize()); // null-stream memset vs. launches on the non-blocking stream
    return THX_OK;
}

size_t thx_time_queue_bytes(uint32_t hit_capacity, uint32_t store_object_id) {
    return 4 + (store_object_id ? 2 : 1) * 4 * (size_t)hit_capacity; // response.time.store.glsl:4-10
}

size_t thx_hit_queue_bytes(uint32_t hit_capacity, uint32_t hit_polarized) {
    const size_t columns = hit_polarized ? 20 : 13; // response.queue.glsl:6-33
    return 4 + columns * 4 * (size_t)hit_capacity;
}

// the counter first, then of every column only the entries that were appended (a queue is sized for the worst case,
// maxHitsPerThread * batch: reading it whole would move ~10x more); `s`: the stream the copies run on (its earlier work is done)
static int copyHitQueue(thx_tracer* t, uint32_t* queue, void* out, size_t nbytes, int clear, hipStream_t s) {
    const size_t cap = t->desc.hit_capacity, columns = hitColumns(t->desc);
    if (nbytes < 4) return fail(THX_ERR_INVALID, "output buffer too small");
    uint32_t count = 0;
    THX_HIP(hipMemcpyAsync(&count, queue, 4, hipMemcpyDeviceToHost, s));
    THX_HIP(hipStreamSynchronize(s));
    std::memcpy(out, &count, 4);
    const size_t n = std::min<size_t>(count, cap);
    for (size_t c = 0; c < columns && n > 0; ++c) {
        const size_t off = 4 + c * cap * 4;
        if (off >= nbytes) break;
        const size_t bytes = std::min(n * 4, nbytes - off);
        THX_HIP(hipMemcpyAsync((char*)out + off, (const char*)queue + off, bytes, hipMemcpyDeviceToHost, s));
    }
    if (clear) THX_HIP(hipMemsetAsync(queue, 0, 4, s)); // (completed before returning: the next launch into this queue must not race it)
    THX_HIP(hipStreamSynchronize(s));
    return THX_OK;
}

int thx_tracer_read_hits(thx_tracer* t, void* out, size_t nbytes, int clear) {
    if (!t || !out) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (!t->hits) return fail(THX_ERR_INVALID, "tracer was not created with the record response");
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize());
    if (int rc = deviceErrorCheck(t)) return rc;
    return copyHitQueue(t, (t->hits1 && t->lastHitCfg == 1) ? t->hits1 : t->hits, out, nbytes, clear, t->ctx->stream);
}

int thx_tracer_read_hits_cfg(thx_tracer* t, uint32_t cfg, void* out, size_t nbytes, int clear) {
    if (!t || !out) return fail(THX_ERR_INVALID, "null argument");
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> guard(t->ctx->lock);
        if (!t->hits1) return fail(THX_ERR_INVALID, "tracer has no hit queue per pipeline config (record / store-time response of a tracing kind)");
        if (cfg > 1) return fail(THX_ERR_INVALID, "config index must be 0 or 1");
        ev = t->evHits[cfg];
        if (!t->ctx->copyStream) {
            THX_HIP(hipSetDevice(t->ctx->device));
            THX_HIP(hipStreamCreateWithFlags(&t->ctx->copyStream, hipStreamNonBlocking));
        }
    }
    // waits for THIS config's launch only -- outside the context's lock: the other config's batch is launched and runs meanwhile
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipEventSynchronize(ev));
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    uint32_t bits = 0; // (the error word as of now: an error of the batch still running is reported by that batch's read)
    THX_HIP(hipMemcpyAsync(&bits, t->devError, 4, hipMemcpyDeviceToHost, t->ctx->copyStream));
    THX_HIP(hipStreamSynchronize(t->ctx->copyStream));
    if (bits != 0) {
        THX_HIP(hipDeviceSynchronize());
        return deviceErrorCheck(t) == THX_OK ? fail(THX_ERR_DEVICE, "device error (already reported)") : THX_ERR_DEVICE;
    }
    return copyHitQueue(t, cfg == 1 ? t->hits1 : t->hits, out, nbytes, clear, t->ctx->copyStream);
}

static int writeQueue(thx_tracer* t, float* dst, size_t have, const void* src, size_t nbytes, const char* what) {
    if (!dst) return fail(THX_ERR_INVALID, std::string("tracer was created without a ") + what);
    if (nbytes != have) return fail(THX_ERR_INVALID, std::string("size of the ") + what + " queue does not match its capacity / polarization");
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, t->ctx->stream));
    THX_HIP(hipStreamSynchronize(t->ctx->stream)); // `src` may be reused by the caller
    return THX_OK;
}

int thx_tracer_write_light(thx_tracer* t, uint32_t cfg, const void* src, size_t nbytes) {
    if (!t || !src || cfg > 1) return fail(THX_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    const size_t have = ((t->desc.flags & THX_FLAG_LIGHT_QUEUE_POLARIZED) ? 15 : 8) * 4 * (size_t)t->desc.light_queue_capacity;
    return writeQueue(t, t->lightIn[cfg], have, src, nbytes, "host light source");
}

int thx_tracer_write_wavelengths(thx_tracer* t, uint32_t cfg, const void* src, size_t nbytes) {
    if (!t || !src || cfg > 1) return fail(THX_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    return writeQueue(t, t->lamIn[cfg], 2 * 4 * (size_t)t->desc.wavelength_queue_capacity, src, nbytes, "host wavelength source");
}

int thx_tracer_read_samples(thx_tracer* t, void* light_out, size_t light_nbytes, void* wavelength_out, size_t wavelength_nbytes) {
    if (!t || !light_out || !wavelength_out) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (t->desc.kind != THX_TRACER_LIGHT_SAMPLER) return fail(THX_ERR_INVALID, "tracer is not a THX_TRACER_LIGHT_SAMPLER");
    const size_t nb = 4 + hitColumns(t->desc) * 4 * (size_t)t->desc.capacity, nl = 4 + 2 * 4 * (size_t)t->desc.capacity;
    if (light_nbytes != nb || wavelength_nbytes != nl) return fail(THX_ERR_INVALID, "queue sizes do not match the sampler's capacity / polarization");
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipMemcpyAsync(light_out, t->hits, nb, hipMemcpyDeviceToHost, t->ctx->stream));
    THX_HIP(hipMemcpyAsync(wavelength_out, t->lamOut, nl, hipMemcpyDeviceToHost, t->ctx->stream));
    THX_HIP(hipStreamSynchronize(t->ctx->stream));
    return THX_OK;
}

int thx_tracer_write_hits(thx_tracer* t, const void* src, size_t nbytes) {
    if (!t || !src) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (!t->replayIn) return fail(THX_ERR_INVALID, "tracer is not a THX_TRACER_HIT_REPLAY");
    const size_t have = thx_hit_queue_bytes(t->desc.capacity, (t->desc.flags & THX_FLAG_POLARIZED) ? 1u : 0u);
    if (nbytes != have) return fail(THX_ERR_INVALID, "queue size does not match the replay's capacity / polarization");
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipMemcpyAsync(t->replayIn, src, nbytes, hipMemcpyHostToDevice, t->ctx->stream));
    THX_HIP(hipStreamSynchronize(t->ctx->stream)); // `src` may be reused by the caller
    return THX_OK;
}

int thx_tracer_hits_device_ptr(thx_tracer* t, uint64_t* dev_addr) {
    if (!t || !dev_addr) return fail(THX_ERR_INVALID, "null argument");
    *dev_addr = (uint64_t)(uintptr_t)t->hits;
    return THX_OK;
}

int thx_tracer_read_track(thx_tracer* t, float* out, size_t n_floats) {
    if (!t || !out) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (!t->track) return fail(THX_ERR_INVALID, "tracer was created without THX_FLAG_TRACK");
    size_t have = (size_t)THX_TRACK_WORDS * t->trackEvents * t->desc.capacity;
    if (n_floats > have) return fail(THX_ERR_INVALID, "track buffer is smaller than requested");
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipDeviceSynchronize());
    if (int rc = deviceErrorCheck(t)) return rc;
    THX_HIP(hipMemcpy(out, t->track, sizeof(float) * n_floats, hipMemcpyDeviceToHost));
    return THX_OK;
}

int thx_tracer_last_kernel_ms(thx_tracer* t, uint32_t cfg, float* ms) {
    if (!t || !ms || cfg >= THX_MAX_SLOTS) return fail(THX_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    if (t->lastSlot[cfg] < 0) return fail(THX_ERR_INVALID, "no launch recorded for this config");
    THX_HIP(hipSetDevice(t->ctx->device));
    THX_HIP(hipEventSynchronize(t->evStop[t->lastSlot[cfg]]));
    THX_HIP(hipEventElapsedTime(ms, t->evStart[t->lastSlot[cfg]], t->evStop[t->lastSlot[cfg]]));
    return THX_OK;
}

int thx_tracer_kernel_times(thx_tracer* t, float* ms_out, uint32_t n, uint32_t* n_written) {
    if (!t || !ms_out || !n_written) return fail(THX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(t->ctx->lock);
    THX_HIP(hipSetDevice(t->ctx->device));
    uint64_t have = std::min<uint64_t>(t->launches, thx_tracer::kRing);
    uint32_t take = (uint32_t)std::min<uint64_t>(have, n);
    for (uint32_t k = 0; k < take; ++k) { // oldest of the requested first
        uint32_t slot = (uint32_t)((t->launches - take + k) % thx_tracer::kRing);
        THX_HIP(hipEventSynchronize(t->evStop[slot]));
        THX_HIP(hipEventElapsedTime(&ms_out[k], t->evStart[slot], t->evStop[slot]));
    }
    *n_written = take;
    return THX_OK;
}

static int probe(thx_ctx* ctx, bool guide, const thx_philox_params* philox, uint32_t flags, const thx_target_params* target,
                 const thx_guide_params* g, const float* lo, const float* hi, uint32_t n, float* out) {
    if (!ctx || !philox || !lo || !hi || !out) return fail(THX_ERR_INVALID, "null argument");
    if (n == 0) return THX_OK;
    std::lock_guard<std::mutex> guard(ctx->lock);
    THX_HIP(hipSetDevice(ctx->device));
    const size_t words = guide ? 13 : 25;
    float* d = nullptr;
    THX_HIP(hipMalloc((void**)&d, 4 * words * (size_t)n));
    LaunchArgs a;
    std::memset(&a, 0, sizeof(a));
    a.p.philox = *philox;
    a.flags = flags;
    if (target) a.p.target = *target;
    if (g) a.p.guide = *g;
    hipError_t e = launchProbe(guide, a, lo, hi, n, d, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d, 4 * words * (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hipFail(e, "probe");
    return THX_OK;
}

int thx_probe_target(thx_ctx* ctx, const thx_philox_params* philox, uint32_t flags, const thx_target_params* target,
                     const float box_min[3], const float box_max[3], uint32_t n, float* target_out) {
    if (!target) return fail(THX_ERR_INVALID, "null argument");
    return probe(ctx, false, philox, flags, target, nullptr, box_min, box_max, n, target_out);
}

int thx_probe_guide(thx_ctx* ctx, const thx_philox_params* philox, uint32_t flags, const thx_guide_params* guide,
                    const float box_min[3], const float box_max[3], uint32_t n, float* guide_out) {
    if (!guide) return fail(THX_ERR_INVALID, "null argument");
    return probe(ctx, true, philox, flags, nullptr, guide, box_min, box_max, n, guide_out);
}

int thx_probe_unit(thx_ctx* ctx, uint32_t which, uint64_t addr, const float* in, uint32_t n_in, uint32_t n, float* out, uint32_t n_out) {
    if (!ctx || !in || !out) return fail(THX_ERR_INVALID, "null argument");
    static const uint32_t cols[THX_UNIT_COUNT][2] = {{2, 1}, {1, 4}, {5, 3}, {3, 2}, {6, 1}, {5, 4}, {9, 5}, {9, 2}, {7, 4}, {6, 4}, {1, 1},
                                                     {4, 1}, {6, 2}, {4, 1}, {2, 3}};
    if (which >= THX_UNIT_COUNT) return fail(THX_ERR_INVALID, "unknown unit probe");
    if (n_in != cols[which][0] || n_out != cols[which][1]) return fail(THX_ERR_INVALID, "column counts do not match the probed function");
    if (n == 0) return THX_OK;
    std::lock_guard<std::mutex> guard(ctx->lock);
    THX_HIP(hipSetDevice(ctx->device));
    float *d_in = nullptr, *d_out = nullptr;
    THX_HIP(hipMalloc((void**)&d_in, 4 * (size_t)n_in * n));
    hipError_t e = hipMalloc((void**)&d_out, 4 * (size_t)n_out * n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, 4 * (size_t)n_in * n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launchUnitProbe(which, addr, d_in, n_in, n, d_out, n_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, 4 * (size_t)n_out * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return hipFail(e, "thx_probe_unit");
    return THX_OK;
}

int thx_rng_fill(thx_ctx* ctx, const thx_philox_params* p, uint32_t base_stream, uint32_t n_streams,
                 uint32_t first_dim, uint32_t n_samples, uint32_t* out_u32, float* out_f32) {
    if (!ctx || !p) return fail(THX_ERR_INVALID, "null argument");
    size_t total = (size_t)n_streams * n_samples;
    if (total == 0) return THX_OK;
    THX_HIP(hipSetDevice(ctx->device));
    uint32_t* d_u = nullptr;
    float* d_f = nullptr;
    hipError_t e = hipSuccess;
    if (out_u32) e = hipMalloc((void**)&d_u, 4 * total);
    if (e == hipSuccess && out_f32) e = hipMalloc((void**)&d_f, 4 * total);
    if (e == hipSuccess) e = launchRngFill(*p, base_stream, n_streams, first_dim, n_samples, d_u, d_f, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && out_u32) e = hipMemcpy(out_u32, d_u, 4 * total, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_f32) e = hipMemcpy(out_f32, d_f, 4 * total, hipMemcpyDeviceToHost);
    if (d_u) (void)hipFree(d_u);
    if (d_f) (void)hipFree(d_f);
    if (e != hipSuccess) return hipFail(e, "thx_rng_fill");
    return THX_OK;
}

int thx_rng_fill_sobol(thx_ctx* ctx, uint32_t seed, uint32_t base_stream, uint32_t n_streams, uint32_t first_dim,
                       uint32_t n_samples, uint32_t sample_dim, float* out_f32) {
    if (!ctx || !out_f32) return fail(THX_ERR_INVALID, "null argument");
    if (sample_dim != 1 && sample_dim != 2) return fail(THX_ERR_INVALID, "sample_dim must be 1 or 2");
    const size_t total = (size_t)n_streams * n_samples * sample_dim;
    if (total == 0) return THX_OK;
    THX_HIP(hipSetDevice(ctx->device));
    float* d = nullptr;
    THX_HIP(hipMalloc((void**)&d, 4 * total));
    hipError_t e = launchSobolFill(seed, base_stream, n_streams, first_dim, n_samples, sample_dim, d, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out_f32, d, 4 * total, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hipFail(e, "thx_rng_fill_sobol");
    return THX_OK;
}

int thx_reduce_histograms(thx_ctx* ctx, const float* hists, uint32_t n_hist, uint32_t n_bins, float norm, float* out) {
    if (!ctx || !out || (!hists && n_hist)) return fail(THX_ERR_INVALID, "null argument");
    if (n_bins == 0) return THX_OK;
    std::lock_guard<std::mutex> guard(ctx->lock);
    THX_HIP(hipSetDevice(ctx->device));
    float *d_in = nullptr, *d_out = nullptr;
    const size_t nb = sizeof(float) * (size_t)n_hist * n_bins;
    THX_HIP(hipMalloc((void**)&d_in, nb ? nb : 16));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(float) * n_bins);
    if (e == hipSuccess && nb) e = hipMemcpyAsync(d_in, hists, nb, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launchReduce(d_in, d_out, n_bins, n_hist, norm, nullptr, nullptr, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(float) * n_bins, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return hipFail(e, "thx_reduce_histograms");
    return THX_OK;
}

int thx_tracer_stats_device_ptr(thx_tracer* t, uint64_t* dev_addr) {
    if (!t || !dev_addr) return fail(THX_ERR_INVALID, "null argument");
    *dev_addr = (uint64_t)(uintptr_t)t->stats;
    return THX_OK;
}

} // extern "C"

// ------------------------------------------------------------------ RCCL
// The one exchange step of the sharded path: sum of the per-GPU time histograms. RCCL is loaded on
// first use (dlopen), so single-GPU users never map it.
namespace {
struct Id128 { char b[128]; }; // ncclUniqueId: passed by value
struct RcclApi {
    void* lib = nullptr;
    int (*getUniqueId)(void*) = nullptr;
    int (*commInitRank)(void**, int, Id128, int) = nullptr;
    int (*commDestroy)(void*) = nullptr;
    int (*allReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*allGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*getErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rcclLock;
bool loadRccl(std::string& why) {
    std::lock_guard<std::mutex> guard(g_rcclLock);
    if (g_rccl.lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { why = std::string("cannot load librccl: ") + dlerror(); return false; }
    g_rccl.getUniqueId = reinterpret_cast<decltype(g_rccl.getUniqueId)>(dlsym(g_rccl.lib, "ncclGetUniqueId"));
    g_rccl.commInitRank = reinterpret_cast<decltype(g_rccl.commInitRank)>(dlsym(g_rccl.lib, "ncclCommInitRank"));
    g_rccl.commDestroy = reinterpret_cast<decltype(g_rccl.commDestroy)>(dlsym(g_rccl.lib, "ncclCommDestroy"));
    g_rccl.allReduce = reinterpret_cast<decltype(g_rccl.allReduce)>(dlsym(g_rccl.lib, "ncclAllReduce"));
    g_rccl.allGather = reinterpret_cast<decltype(g_rccl.allGather)>(dlsym(g_rccl.lib, "ncclAllGather"));
    g_rccl.getErrorString = reinterpret_cast<decltype(g_rccl.getErrorString)>(dlsym(g_rccl.lib, "ncclGetErrorString"));
    if (!g_rccl.getUniqueId || !g_rccl.commInitRank || !g_rccl.commDestroy || !g_rccl.allReduce || !g_rccl.allGather) {
        why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclAllGather";
        dlclose(g_rccl.lib);
        g_rccl = RcclApi{};
        return false;
    }
    return true;
}
int rcclFail(int rc, const char* what) {
    return fail(THX_ERR_HIP, std::string(what) + ": " + (g_rccl.getErrorString ? g_rccl.getErrorString(rc) : "RCCL error ") + " (" + std::to_string(rc) + ")");
}
} // namespace

struct thx_comm {
    thx_ctx* ctx = nullptr;
    void* nccl = nullptr;
    int rank = 0, world = 1;
    float* gathered = nullptr; // [world][n] staging of thx_hist_gather_reduce
    size_t gatheredFloats = 0;
};

extern "C" {

int thx_comm_unique_id(uint8_t out[THX_COMM_ID_BYTES]) {
    if (!out) return fail(THX_ERR_INVALID, "null argument");
    std::string why;
    if (!loadRccl(why)) return fail(THX_ERR_UNSUPPORTED, why);
    static_assert(THX_COMM_ID_BYTES == 128, "ncclUniqueId is 128 bytes");
    int rc = g_rccl.getUniqueId(out);
    return rc == 0 ? THX_OK : rcclFail(rc, "ncclGetUniqueId");
}

int thx_comm_init(thx_ctx* ctx, const uint8_t unique_id[THX_COMM_ID_BYTES], int rank, int world_size, thx_comm** out) {
    if (!ctx || !unique_id || !out) return fail(THX_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(THX_ERR_INVALID, "rank / world size out of range");
    std::string why;
    if (!loadRccl(why)) return fail(THX_ERR_UNSUPPORTED, why);
    THX_HIP(hipSetDevice(ctx->device));
    Id128 id;
    std::memcpy(id.b, unique_id, 128);
    void* comm = nullptr;
    int rc = g_rccl.commInitRank(&comm, world_size, id, rank);
    if (rc != 0) return rcclFail(rc, "ncclCommInitRank");
    thx_comm* c = new (std::nothrow) thx_comm();
    if (!c) { g_rccl.commDestroy(comm); return fail(THX_ERR_INVALID, "out of host memory"); }
    c->ctx = ctx; c->nccl = comm; c->rank = rank; c->world = world_size;
    *out = c;
    return THX_OK;
}

int thx_comm_destroy(thx_comm* comm) {
    if (!comm) return THX_OK;
    (void)hipSetDevice(comm->ctx->device);
    if (comm->nccl) g_rccl.commDestroy(comm->nccl);
    if (comm->gathered) (void)hipFree(comm->gathered);
    delete comm;
    return THX_OK;
}

int thx_hist_allreduce(thx_comm* comm, uint64_t d_hist, size_t n_floats, void* stream) {
    if (!comm || (d_hist == 0 && n_floats)) return fail(THX_ERR_INVALID, "null argument");
    if (n_floats == 0) return THX_OK;
    THX_HIP(hipSetDevice(comm->ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : comm->ctx->stream;
    void* p = (void*)(uintptr_t)d_hist;
    // ncclFloat32 = 7, ncclSum = 0 (rccl.h); in place
    int rc = g_rccl.allReduce(p, p, n_floats, 7, 0, comm->nccl, s);
    return rc == 0 ? THX_OK : rcclFail(rc, "ncclAllReduce");
}

int thx_stats_allreduce(thx_comm* comm, uint64_t d_counters, size_t n, void* stream) {
    if (!comm || (d_counters == 0 && n)) return fail(THX_ERR_INVALID, "null argument");
    if (n == 0) return THX_OK;
    THX_HIP(hipSetDevice(comm->ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : comm->ctx->stream;
    void* p = (void*)(uintptr_t)d_counters;
    // ncclUint64 = 5, ncclSum = 0 (rccl.h); in place
    int rc = g_rccl.allReduce(p, p, n, 5, 0, comm->nccl, s);
    return rc == 0 ? THX_OK : rcclFail(rc, "ncclAllReduce (u64)");
}

int thx_hist_gather_reduce(thx_comm* comm, uint64_t d_hist, size_t n_floats, void* stream) {
    if (!comm || (d_hist == 0 && n_floats)) return fail(THX_ERR_INVALID, "null argument");
    if (n_floats == 0) return THX_OK;
    if (n_floats > 0xFFFFFFFFull) return fail(THX_ERR_INVALID, "histogram too large");
    THX_HIP(hipSetDevice(comm->ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : comm->ctx->stream;
    const size_t need = n_floats * (size_t)comm->world;
    if (need > comm->gatheredFloats) {
        THX_HIP(hipStreamSynchronize(s));
        if (comm->gathered) (void)hipFree(comm->gathered);
        comm->gathered = nullptr; comm->gatheredFloats = 0;
        THX_HIP(hipMalloc((void**)&comm->gathered, sizeof(float) * need));
        comm->gatheredFloats = need;
    }
    // every rank's curve, in rank order, on every rank ...
    int rc = g_rccl.allGather((const void*)(uintptr_t)d_hist, comm->gathered, n_floats, 7, comm->nccl, s);
    if (rc != 0) return rcclFail(rc, "ncclAllGather");
    // ... summed by the slab reducer: Kahan partials per lane, fixed butterfly -- the same bits on every rank and in every run
    THX_HIP(launchReduce(comm->gathered, (float*)(uintptr_t)d_hist, (uint32_t)n_floats, (uint32_t)comm->world, 1.0f, nullptr, nullptr, s));
    return THX_OK;
}

} // extern "C"
