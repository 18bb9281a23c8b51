/*
 * theia_hip.h -- C ABI of libtheia_hip.so: the MI355X (gfx950) transient
 * photon-tracing backend behind theia's SceneForwardTracer / VolumeForwardTracer.
 *
 * The reference (tkerscher/theia) has no FFI; the seam this library sits
 * behind is the `hephaistos` PipelineStage protocol. Each
 * entry point below names the reference interface it replaces (paths relative
 * to /root/reference). Conventions: C linkage, opaque handles, `int` status
 * (0 ok, <0 error; thx_last_error() gives the text), caller owns host memory,
 * the library owns device memory, calls on one context are serialised by the
 * caller, launches are asynchronous on a HIP stream, reads synchronise.
 *
 * Parameter blocks are the reference's UBO layouts verbatim (scalar layout,
 * tightly packed, docs/pipeline/extend.md:163-164).
 */
#ifndef THEIA_HIP_H
#define THEIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THX_OK 0
#define THX_ERR_INVALID (-1)     /* bad argument */
#define THX_ERR_HIP (-2)         /* HIP runtime error */
#define THX_ERR_NO_DEVICE (-3)   /* no usable gfx950 device */
#define THX_ERR_UNSUPPORTED (-4) /* component outside the supported closed set */
#define THX_ERR_DEVICE (-5)      /* a kernel reported an error through the tracer's device error word (THX_DEVERR_*) */
#define THX_DEVERR_WATCHDOG 1u   /* the staged kernel's scheduler watchdog tripped: the batch is incomplete */

/* Slots. Every call that takes a `cfg` / `slot` index addresses one of THX_MAX_SLOTS result slots of a tracer (partial
 * histograms, reduced curve, readback buffer). Slots 0 and 1 are the two pipeline configs of the reference
 * (docs/pipeline/pipeline.md:32-38); a host that keeps more than two batches in flight (theia_amd.hp.pipeline.PipelineScheduler
 * does, behind the two-config facade) uses the others as well. Slots beyond 1 are allocated on first use. */
#define THX_MAX_SLOTS 16

typedef struct thx_ctx thx_ctx;
typedef struct thx_scene thx_scene;
typedef struct thx_tracer thx_tracer;
typedef struct thx_comm thx_comm;

#pragma pack(push, 4)

/* src/theia/trace.py:354-360 (Tracer.DispatchParams) */
typedef struct {
    uint32_t batchSize;
    uint32_t groupCountX, groupCountY, groupCountZ; /* kept for layout parity; unused */
} thx_dispatch_params;

/* src/theia/random.py:253-254; shader/random.philox.glsl:8-11 */
typedef struct {
    uint32_t key[2];
    uint32_t offset[4];
} thx_philox_params;

/* The Cherenkov family of light sources (THX_FLAG_LIGHT_KIND_* below; forward mode: the `sampleLight(wavelength, medium, idx, dim)`
 * of each shader). Their parameter blocks are the reference's uniform blocks, field by field: */
/* CherenkovLightSource: src/theia/light.py:1219-1228; shader/lightsource.cherenkov.simple.glsl:8-18 */
typedef struct {
    float trackStart[3];
    float startTime;
    float trackEnd[3];
    float endTime;
    float trackDir[3]; /* unit, end - start (light.py:1262-1268 _finishParams) */
    float trackDist;
} thx_cherenkov_params;
/* CherenkovTrackLightSource: light.py:1376-1377; shader/lightsource.cherenkov.track.glsl:9-21. `track` = device address (lo, hi)
 * of a ParticleTrack: uint32 trackLength (number of SEGMENTS), then trackLength + 1 vertices {x, y, z, t} (light.py:1285-1297) */
typedef struct {
    uint32_t track[2];
} thx_cherenkov_track_params;
/* MuonTrackLightSource: light.py:1456-1465; shader/lightsource.particles.muon.glsl:13-27 */
typedef struct {
    float startPosition[3];
    float startTime;
    float endPosition[3];
    float endTime;
    float energyScale;
    float a_angular, b_angular;
} thx_muon_track_params;
/* ParticleCascadeLightSource: light.py:1591-1601; shader/lightsource.particles.cascade.glsl:14-39 */
typedef struct {
    float startPosition[3];
    float startTime;
    float direction[3];
    float effectiveLength;
    float a_angular, b_angular;
    float a_long, b_long;
} thx_cascade_params;

/* One block for whichever light source the tracer was created with.
 * SphericalLightSource: src/theia/light.py:1132-1139; shader/lightsource.spherical.glsl:7-14.
 * PencilLightSource (THX_FLAG_LIGHT_PENCIL): light.py:1065-1073; shader/lightsource.pencil.glsl:4-16
 * -- contribFwd carries its budget, direction / stokes / polRef are read by it and the cone only.
 * ConeLightSource (THX_FLAG_LIGHT_CONE): light.py:935-945; shader/lightsource.cone.glsl:6-20.
 * Cherenkov family: the member named after the source. */
typedef union {
    struct {
        float position[3];
        float contribFwd;
        float contribBwd;
        float t_min, t_max;
        float direction[3];
        float stokes[4];
        float polRef[3];
        float cosOpeningAngle;
    };
    thx_cherenkov_params cherenkov;
    thx_cherenkov_track_params cherenkovTrack;
    thx_muon_track_params muonTrack;
    thx_cascade_params cascade;
} thx_light_params;

/* src/theia/light.py:272-273 (const) / :308-312 (uniform) / :393-394 (function) */
typedef struct {
    float p0; /* const: lambda | uniform: lam_min */
    float p1; /*               | uniform: lam_max */
    float p2; /*               | uniform: contrib | function: contrib */
    uint32_t table[2]; /* FunctionWavelengthSource: device address of its Table1D (inverse CDF) */
} thx_wavelength_params;

/* src/theia/target.py:105-112; shader/target.sphere.glsl:8-16 */
typedef struct {
    float position[3];
    float radius;
    float invPos[3];
    float invRad;
    float hemisphereProb; /* inner sphere / disk / flat: prob = 1 / area */
    /* DiskTarget / FlatTarget only (THX_FLAG_TARGET_DISK / _FLAT; target.py:252-259, 384-391;
     * target.disk.glsl:6-13, target.flat.glsl:6-15): position = centre, objToWorld column major */
    float normal[3];
    float objToWorld[9];
    float width, height;
} thx_target_params;

/* src/theia/target.py:496-501; shader/target_guide.sphere.glsl:7-10 */
typedef struct {
    float position[3];
    float radius;
    float prob; /* sphere: unused by the shader; disk / flat: 1 / area */
    /* DiskTargetGuide / FlatTargetGuide only (THX_FLAG_GUIDE_DISK / _FLAT; target.py:575-583, 672-679;
     * target_guide.disk.glsl:6-12, target_guide.flat.glsl:6-13) */
    float normal[3];
    float objToWorld[9];
    float width, height;
} thx_guide_params;

/* shader/ray.propagate.glsl:18-26 */
typedef struct {
    float scatterCoefficient;
    float lowerBBoxCorner[3];
    float upperBBoxCorner[3];
    float maxTime;
    float maxDist;
} thx_propagate_params;

/* src/theia/trace.py:589-598; shader/tracer.volume.forward.glsl:18-23 */
typedef struct {
    uint32_t medium[2]; /* device address (lo, hi) */
    int32_t objectId;
    thx_propagate_params propagation;
} thx_volume_trace_params;

/* src/theia/trace.py:1136-1145; shader/tracer.scene.forward.glsl:36-41 */
typedef struct {
    int32_t targetId;
    uint32_t sceneMedium[2];
    thx_propagate_params propagation;
} thx_scene_trace_params;

/* src/theia/response.py:1256-1257; shader/response.histogram.glsl:11-14.
 * THX_FLAG_KERNEL_HISTOGRAM: KernelHistogramHitResponse, response.py:1495-1501,
 * shader/response.histogram.kernel.glsl:11-19 (Gaussian kernel bandwidth and support) */
typedef struct {
    float t0;
    float binSize;
    float kernelBandwidth;
    float kernelSupport;
} thx_response_params;

/* everything one batch needs (one `stage.update(i)` round of the reference's
 * pipeline, docs/pipeline/pipeline.md:32-38) */
typedef struct {
    thx_dispatch_params dispatch;
    uint32_t photon_offset; /* index/RNG stream of the first photon (multi-GPU sharding) */
    thx_philox_params philox;
    thx_light_params light;
    thx_wavelength_params wavelength;
    thx_response_params response;
    thx_volume_trace_params volume; /* kind == THX_TRACER_VOLUME_FORWARD */
    thx_target_params target;
    thx_scene_trace_params scene;   /* kind == THX_TRACER_SCENE_FORWARD */
    thx_guide_params guide;
    /* THX_FLAG_PHOTON: the reference runs nRuns dispatches of steps_per_run steps and starts
     * run r at RNG dimension r * rng_per_run (trace.py:2489-2491, 2655-2659) */
    uint32_t photon_steps_per_run;
    uint32_t photon_rng_per_run;
} thx_run_params;

#pragma pack(pop)

/* THX_TRACER_HIT_REPLAY: HitReplay (response.py:278-423, shader/response.replay.glsl): no tracing,
 * every item of an input hit queue of `capacity` entries (uploaded with thx_tracer_write_hits;
 * THX_FLAG_POLARIZED = the queue has the stokes / polRef columns) is fed to the response with
 * idx = its position in the queue and RNG dimension 0 */
enum { THX_TRACER_VOLUME_FORWARD = 0, THX_TRACER_SCENE_FORWARD = 1, THX_TRACER_HIT_REPLAY = 2, THX_TRACER_LIGHT_SAMPLER = 3 };
/* THX_TRACER_LIGHT_SAMPLER: LightSampler (light.py:492-689, shader/lightsource.sample.glsl): no
 * tracing, sample idx + photon_offset (its `baseCount`) of the wavelength and light source selected by
 * the flags is written to item idx of two queues of `capacity` entries (dispatch.batchSize = its
 * `count`, volume.medium = its `medium`; THX_FLAG_POLARIZED = POLARIZATION and LIGHT_QUEUE_POLARIZED
 * of the output); read with thx_tracer_read_samples */

/* compile-time macros of the reference's shaders (trace.py:660-666, 1226-1235) */
enum {
    THX_FLAG_POLARIZED             = 1u << 0,
    THX_FLAG_DISABLE_DIRECT        = 1u << 1,
    THX_FLAG_DISABLE_MIS           = 1u << 2,
    THX_FLAG_DISABLE_TRANSMISSION  = 1u << 3,
    THX_FLAG_DISABLE_VOLUME_BORDER = 1u << 4,
    THX_FLAG_HIT_USE_TRANSMITTED   = 1u << 5,
    THX_FLAG_WAVELENGTH_UNIFORM    = 1u << 6, /* UniformWavelengthSource, else Const */
    THX_FLAG_STATISTICS            = 1u << 8, /* EventStatisticCallback + traversal counters */
    THX_FLAG_TRACK                 = 1u << 9, /* per-event state dump (parity probe, TrackRecordCallback) */
    THX_FLAG_LIGHT_PENCIL          = 1u << 10, /* PencilLightSource, else SphericalLightSource */
    THX_FLAG_KERNEL_HISTOGRAM      = 1u << 13, /* histogram response smears every hit with a Gaussian kernel */
    THX_FLAG_LIGHT_HOST            = 1u << 14, /* HostLightSource (light.py:692-772, lightsource.host.glsl): photon k of a
                                                  batch starts as item k of the queue uploaded with thx_tracer_write_light */
    THX_FLAG_LIGHT_QUEUE_POLARIZED = 1u << 15, /* LIGHT_QUEUE_POLARIZED: that queue has the stokes / polRef columns */
    THX_FLAG_WAVELENGTH_HOST       = 1u << 16, /* HostWavelengthSource (light.py:87-150): thx_tracer_write_wavelengths */
    THX_FLAG_LIGHT_CONE            = 1u << 17, /* ConeLightSource (lightsource.cone.glsl) */
    THX_FLAG_LIGHT_POLARIZED       = 1u << 18, /* LIGHTSOURCE_POLARIZED of the cone source */
    THX_FLAG_WAVELENGTH_FUNCTION   = 1u << 19, /* FunctionWavelengthSource (wavelengthsource.function.glsl) */
    THX_FLAG_TARGET_DISK           = 1u << 20, /* volume tracer: DiskTarget (target.disk.glsl) */
    THX_FLAG_TARGET_FLAT           = 1u << 21, /* volume tracer: FlatTarget (target.flat.glsl) */
    THX_FLAG_GUIDE_DISK            = 1u << 22, /* scene tracer: DiskTargetGuide (target_guide.disk.glsl), else sphere */
    THX_FLAG_GUIDE_FLAT            = 1u << 23, /* scene tracer: FlatTargetGuide (target_guide.flat.glsl) */
    THX_FLAG_RNG_SOBOL             = 1u << 24, /* SobolQRNG (random.py:285-352, random.sobol.glsl) instead of PhiloxRNG;
                                                  its seed travels in thx_philox_params.key[0] (`offset` is unused
                                                  by the reference's shader as well) */
    THX_FLAG_TARGET_MISS_INFINITE  = 1u << 25, /* DiskTarget / FlatTarget: report a missed outline (or a plane behind the ray) as infinitely
                                                far. Default (flag clear) = the reference: target.disk.glsl:60 / target.flat.glsl:66 return the
                                                distance to the plane regardless, and tracer.volume.forward.glsl:167 stops the ray there */
    /* bits 26-28: light sources of the Cherenkov family (0: none of them -- the LIGHT_* bits above decide) */
    THX_FLAG_LIGHT_KIND_SHIFT      = 26,
    THX_FLAG_LIGHT_KIND_MASK       = 7u << 26,
    THX_FLAG_LIGHT_CHERENKOV       = 1u << 26, /* CherenkovLightSource (lightsource.cherenkov.simple.glsl) */
    THX_FLAG_LIGHT_CHERENKOV_TRACK = 2u << 26, /* CherenkovTrackLightSource (lightsource.cherenkov.track.glsl) */
    THX_FLAG_LIGHT_MUON_TRACK      = 3u << 26, /* MuonTrackLightSource (lightsource.particles.muon.glsl) */
    THX_FLAG_LIGHT_CASCADE         = 4u << 26, /* ParticleCascadeLightSource (lightsource.particles.cascade.glsl) */
    THX_FLAG_FRANK_TAMM_PHOTONS    = 1u << 29, /* FRANK_TAMM_USE_PHOTON_COUNT of the two Cherenkov sources: radiance in photons, else eV */
    THX_FLAG_FRANK_TAMM_IS         = 1u << 30, /* FRANK_TAMM_IS of the muon track / cascade sources: the Frank-Tamm factor is NOT applied
                                                  (the wavelength source is taken to have importance sampled it) */
    THX_FLAG_SOURCES_PRESAMPLED    = 1u << 7,  /* set by the library, never by the caller: a tracer with a Cherenkov-family source reads
                                                  light and wavelength samples (and the RNG dimension behind them) from queues its
                                                  sampler kernel filled right before the launch */
    THX_FLAG_PHOTON                = 1u << 12, /* ScenePhotonTracer (trace.py:2370-2665): contribution = survival chance,
                                                  absorption roulette after every step, RNG dimension restarted per run */
    THX_FLAG_TARGET_INNER          = 1u << 11  /* volume tracer: InnerSphereTarget (target.py:142-199), else SphereTarget;
                                                  thx_target_params.hemisphereProb then holds its `prob` = 1/area */
};

/* number of u64 counters returned by thx_tracer_read_stats: 12 event counters
 * in the order of shader/callback.stat.glsl:7-20, then queries, nodes,
 * triangles, instances, hits, lut_reads, responses, steps */
#define THX_N_STATS 20
/* words per event in the track buffer: code, pos.xyz, dir.xyz, time, lin, log, wavelength, dim,
 * stokes.iquv, polRef.xyz ((1,0,0,0), 0 for an unpolarized tracer: callback.track.glsl:84-88), pad */
#define THX_TRACK_WORDS 20

typedef struct {
    uint32_t kind;        /* THX_TRACER_* */
    uint32_t flags;       /* THX_FLAG_* */
    uint32_t path_length; /* PATH_LENGTH macro */
    uint32_t block_size;  /* reference BLOCK_SIZE; accepted for API parity */
    uint32_t n_bins;      /* N_BINS macro (histogram response) */
    uint32_t capacity;    /* max batch size */
    uint32_t response_kind; /* THX_RESPONSE_* */
    uint32_t hit_capacity;  /* HIT_QUEUE_SIZE macro (record response), response.py:261-266 */
    uint32_t hit_polarized; /* HIT_QUEUE_POLARIZED macro */
    uint32_t light_queue_capacity;      /* LIGHT_QUEUE_SIZE of a HostLightSource (0: none) */
    uint32_t wavelength_queue_capacity; /* PHOTON_QUEUE_SIZE of a HostWavelengthSource (0: none) */
    uint32_t reserved[1];
} thx_tracer_desc;

/* which `response(HitItem)` the tracer is built with:
 * HISTOGRAM = HistogramHitResponse(UniformValueResponse()) (response.histogram.glsl),
 * RECORD    = HitRecorder (response.record.glsl): hits appended to a queue with the layout
 *             of shader/response.queue.glsl:6-33 -- uint32 count, then one float column of
 *             hit_capacity entries each for posX posY posZ dirX dirY dirZ nrmX nrmY nrmZ
 *             [stokesI stokesQ stokesU stokesV polX polY polZ] wavelength time contrib and
 *             an int32 column objectId. Runs on the instrumented kernel variants. */
/* STORE_TIME = StoreTimeHitResponse(UniformValueResponse()) (response.py:656-800,
 *             shader/response.time.store.glsl): a hit is kept with probability `value`
 *             (one random draw of the photon's stream); queue = uint32 count, float time[cap],
 *             int32 objectId[cap] (the latter if hit_polarized, which here means storeObjectId) */
enum { THX_RESPONSE_HISTOGRAM = 0, THX_RESPONSE_RECORD = 1, THX_RESPONSE_STORE_TIME = 2 };

/* vertices [n][6] f32 = position + normal (shader/scene.types.glsl:6-9),
 * indices [m][3] u32 (:12-14) -- what hp.Mesh holds (scene.py:425-431) */
typedef struct {
    const float* vertices;
    uint32_t n_vertices;
    const uint32_t* indices;
    uint32_t n_triangles;
} thx_mesh_desc;

/* hp.GeometryInstance + Scene.GLSLGeometry (scene.py:589-596, 628-645):
 * object->world 3x4 row-major, customIndex = detectorId, material device address */
typedef struct {
    uint32_t mesh;
    float transform[12];
    int32_t customIndex;
    uint64_t material;
} thx_instance_desc;

typedef struct {
    uint32_t n_nodes, n_triangles, n_instances, max_depth;
    uint64_t bytes_nodes, bytes_triangles, bytes_instances, bytes_total;
    /* bytes the traversal's own LOADS REQUEST per counted unit (the counters of the instrumented kernels): per child box tested
     * (half a 32-B pair record), per triangle tested (three 12-B vertex loads out of a 64-B record -- the cache moves the lines they
     * touch, and the walk also requests the record behind a leaf's first triangle whether or not it belongs to the leaf, which the
     * triangle counter does not see), per instance entered, per hit reconstructed. Requested, not moved. (SURVEY.md 8(d) fixes
     * contract figures of 32 / 48 / 112 / 132 B for the north star's roofline; bench.py reports both.) */
    uint32_t layout_box_bytes, layout_triangle_bytes, layout_instance_bytes, layout_hit_bytes;
} thx_scene_info;

/* Threading: every call may be made from any thread (each sets its device); calls on tracers and
 * scenes of one context are serialised by a per-context mutex, so a pipeline's update / launch /
 * retrieve threads need no locking of their own. thx_last_error() is per thread. Create and
 * destroy objects while no other thread uses them. */
/* ---- context / device (replaces hp.getCurrentDevice, hp.selectDevice) ---- */
const char* thx_last_error(void);
int thx_device_count(int* out);
int thx_ctx_create(int device, thx_ctx** out);
int thx_ctx_destroy(thx_ctx* ctx);
int thx_ctx_device_name(thx_ctx* ctx, char* buf, size_t n);
int thx_ctx_device_info(thx_ctx* ctx, uint32_t* n_cus, uint32_t* lds_per_cu, uint64_t* hbm_bytes);
int thx_ctx_sync(thx_ctx* ctx);
/* How many of the context's lane streams (thx_tracer_set_concurrency) really run side by side, measured once and remembered: the
 * largest L of 16, 12, 8, 6, 4, 3, 2 for which L idle 200-us kernels, one per lane stream, were all running at the same moment (device
 * timestamps; ~1-3 ms), else 1. Streams that share one of the runtime's hardware queues (GPU_MAX_HW_QUEUES, default 4, read when
 * HIP initialises) run one after the other, and more lanes than run side by side make batches queue up behind each other on a
 * fraction of the chip -- slower than whole-chip launches. The scheduler of theia_amd.hp.pipeline asks before it sets its lanes. */
int thx_ctx_concurrent_kernels(thx_ctx* ctx, uint32_t* out);

/* ---- raw device memory (replaces hp.Tensor/hp.Buffer + hp.updateTensor for the
 * MaterialStore blob; addresses stay plain integers, material.py:1005-1018) ---- */
int thx_mem_alloc(thx_ctx* ctx, size_t nbytes, uint64_t* dev_addr);
int thx_mem_free(thx_ctx* ctx, uint64_t dev_addr);
int thx_mem_upload(thx_ctx* ctx, uint64_t dev_addr, const void* src, size_t nbytes);
int thx_mem_download(thx_ctx* ctx, void* dst, uint64_t dev_addr, size_t nbytes);
/* Page-locks a host buffer of the caller's (the two host buffers of an IOQueue, hephaistos' mapped hp.Buffer): copies between it and the
 * device then run on the DMA engines, beside a kernel that occupies every CU -- a copy to pageable memory is done by a copy KERNEL,
 * which gets no CU while the staged tracing kernel runs (measured: 157 MB of hits per batch read 60 % slower overlapped than one after
 * the other). Registering twice / unregistering what is not registered is not an error. Unregister before freeing the memory. */
int thx_host_register(thx_ctx* ctx, void* ptr, size_t nbytes);
int thx_host_unregister(thx_ctx* ctx, void* ptr);

/* ---- scene (replaces hp.GeometryStore + hp.AccelerationStructure + the
 * Geometries tensor, scene.py:550, 637-668): builds BLAS/TLAS (own BVH2) ---- */
int thx_scene_create(thx_ctx* ctx, const thx_mesh_desc* meshes, uint32_t n_meshes,
                     const thx_instance_desc* instances, uint32_t n_instances, thx_scene** out);
int thx_scene_destroy(thx_scene* scene);
int thx_scene_get_info(const thx_scene* scene, thx_scene_info* out);
/* closest-hit probe: n rays (origin[3], dir[3], tmax) -> out[n][5] =
 * {hit, instance, primitive, t, bary.x, bary.y}[..] as floats (6 per ray) */
int thx_scene_query(thx_scene* scene, const float* rays7, uint32_t n, float* out6);
/* kernel time (ms) of the most recent thx_scene_query (HIP events, second of two runs) */
int thx_scene_query_ms(const thx_scene* scene, float* ms);

/* ---- tracer (replaces hp.Program(compileShader("tracer.*.forward.glsl")) +
 * Tracer.run/_dispatch, trace.py:478-481, 758-766, 1326-1336) ---- */
int thx_tracer_create(thx_ctx* ctx, const thx_tracer_desc* desc, thx_scene* scene_or_null,
                      thx_tracer** out);
int thx_tracer_destroy(thx_tracer* tracer);
/* launches one batch into slot `cfg` (0 / 1 = the pipeline configs, < THX_MAX_SLOTS) on `stream`
 * (hipStream_t; NULL = the stream of the slot's lane, see thx_tracer_set_concurrency). The parameter block is
 * consumed by the call (copied into the kernel arguments): the caller may change it right away. Tracers fed
 * from host queues (THX_FLAG_LIGHT_HOST / _WAVELENGTH_HOST) have device queues for slots 0 and 1 only. */
int thx_tracer_run(thx_tracer* tracer, uint32_t cfg, const thx_run_params* params, void* stream);
/* Batches side by side. A launch of the tracing kernels ends with the life of its last photons -- up to path_length
 * steps one after the other, ~20 us each, while its workgroups run empty (config 2: 0.22 ms per launch whatever the
 * batch size, profiles/r03_launch_tail.txt). With `lanes` > 1, slot s runs on lane s % lanes -- a stream and a share of
 * the tracer's scratch memory of its own -- and a launch takes at most 1 / grid_divisor of the chip's workgroups, so that
 * several batches are in flight at once and the tail of one is covered by the others (the reference's scheduler keeps
 * configs 0 / 1 in flight the same way, docs/pipeline/pipeline.md:74-92). lanes > grid_divisor oversubscribes the chip:
 * workgroups of the next batch start wherever one of an earlier batch ends. (1, 1) (default): every launch takes the
 * whole chip, all slots share the context's stream. Waits for the tracer's work in flight. lanes in [1, THX_MAX_SLOTS],
 * grid_divisor in [1, lanes]; tracers with ONE output queue or track buffer (record / store-time responses, light
 * sampler, hit replay, THX_FLAG_TRACK) accept (1, 1) only. Lanes only run side by side while they have hardware queues of their own
 * (GPU_MAX_HW_QUEUES, read when HIP initialises): see thx_ctx_concurrent_kernels. */
int thx_tracer_set_concurrency(thx_tracer* tracer, uint32_t lanes, uint32_t grid_divisor);
/* HistogramReducer.run (response.py:1188-1197; shader/estimator.reduce.glsl):
 * out[b] = norm * KahanSum_h slab[h][b]; d_out == 0 -> internal buffer of cfg */
int thx_tracer_reduce(thx_tracer* tracer, uint32_t cfg, float norm, uint64_t d_out, void* stream);
/* hp.retrieveTensor + response.result(i) (response.py:1184-1186, 1357-1365) */
int thx_tracer_read_hist(thx_tracer* tracer, uint32_t cfg, float* out, uint32_t n_bins);
/* Asynchronous form of thx_tracer_read_hist: _fetch_hist enqueues the copy of slot `cfg`'s reduced curve and of the device
 * error word into a pinned host buffer of the tracer behind whatever is queued on `stream` (NULL = the slot's lane) and
 * returns; _collect_hist waits for THAT copy only (other slots' batches keep running), performs the checks of
 * thx_tracer_read_hist and hands the curve over. One fetch in flight per slot. (hp.retrieveTensor is a command in the
 * reference's timeline as well; the scheduler's process thread waits for it, docs/pipeline/pipeline.md:80-88.) */
int thx_tracer_fetch_hist(thx_tracer* tracer, uint32_t cfg, void* stream);
int thx_tracer_collect_hist(thx_tracer* tracer, uint32_t cfg, float* out, uint32_t n_bins);
/* HistogramReducer on host arrays (response.py:1065-1197; shader/estimator.reduce.glsl:17-35): out[b] = norm *
 * KahanSum_h hists[h][b] for n_hist histograms of n_bins bins -- the reducer thx_tracer_reduce runs over a tracer's own
 * slabs, on data of the caller's (upload, one reduceKernel launch, download; synchronises) */
int thx_reduce_histograms(thx_ctx* ctx, const float* hists, uint32_t n_hist, uint32_t n_bins, float norm, float* out);
/* Waits for the tracer's launches and returns THX_ERR_DEVICE if a kernel flagged an error in the tracer's
 * device error word (THX_DEVERR_*; the word is cleared). Every thx_tracer_read_* does the same check; call
 * this after thx_tracer_reduce into an EXTERNAL buffer (d_out != 0), which the library never reads back.
 * (The reference counts device errors and never raises, callback.stat.glsl:66-69; a scheduler watchdog has
 * no counterpart there -- it must not pass for a result.) */
int thx_tracer_check(thx_tracer* tracer);
int thx_tracer_hist_device_ptr(thx_tracer* tracer, uint32_t cfg, uint64_t* out);
/* EventStatisticCallback (trace.py:77-187) */
int thx_tracer_read_stats(thx_tracer* tracer, uint64_t* out, uint32_t n);
int thx_tracer_reset_stats(thx_tracer* tracer);
/* device address of the THX_N_STATS u64 counters (for thx_stats_allreduce) */
int thx_tracer_stats_device_ptr(thx_tracer* tracer, uint64_t* dev_addr);
/* development aid (library built with -DTHX_STAGE_PROFILE): per-stage cycles [0..15],
 * rounds [16..31], photons [32..47] of the staged scene kernel since the last call */
int thx_tracer_read_stage_profile(thx_tracer* tracer, uint64_t* out48);
/* HitRecorder queue (replaces hephaistos IOQueue(mode="retrieve") of response.py:219,
 * 268-274): size in bytes of a queue, copy of the device queue to `out` (at most `nbytes`;
 * synchronises), optionally followed by resetting its counter for the next batch. The counter
 * keeps counting past a full queue; entries beyond hit_capacity are dropped. */
size_t thx_hit_queue_bytes(uint32_t hit_capacity, uint32_t hit_polarized);
size_t thx_time_queue_bytes(uint32_t hit_capacity, uint32_t store_object_id); /* THX_RESPONSE_STORE_TIME queue */
int thx_tracer_read_hits(thx_tracer* tracer, void* out, size_t nbytes, int clear);
/* The tracing kinds keep one hit queue per pipeline config (slots 0 / 1; IOQueue's two tensors, response.py:219): _read_hits_cfg waits
 * for the last launch of config `cfg` only and copies ITS queue on a stream of its own, so that the batch of the other config runs while
 * this one is read (thx_tracer_read_hits waits for the device and reads the queue of the most recent launch). */
int thx_tracer_read_hits_cfg(thx_tracer* tracer, uint32_t cfg, void* out, size_t nbytes, int clear);
int thx_tracer_hits_device_ptr(thx_tracer* tracer, uint64_t* dev_addr); /* (the queue of config 0) */
/* HostLightSource / HostWavelengthSource input (replace hephaistos IOQueue(mode="update",
 * skipCounter=True) of light.py:120, 735): copy a host queue without counter -- columns of
 * light_queue_capacity floats posX posY posZ dirX dirY dirZ [stokesI Q U V polX polY polZ]
 * startTime contrib (lightsource.queue.glsl:10-30), resp. wavelength_queue_capacity floats
 * wavelength contrib (wavelengthsource.queue.glsl:10-13) -- to the device queue the next run of
 * pipeline config `cfg` reads */
int thx_tracer_write_light(thx_tracer* tracer, uint32_t cfg, const void* src, size_t nbytes);
int thx_tracer_write_wavelengths(thx_tracer* tracer, uint32_t cfg, const void* src, size_t nbytes);
/* LightSampler output (replaces IOQueue(mode="retrieve") of light.py:607-610): both queues of the
 * last run, each uint32 count + columns of `capacity` entries (light: 8 or 15 columns) */
int thx_tracer_read_samples(thx_tracer* tracer, void* light_out, size_t light_nbytes, void* wavelength_out,
                            size_t wavelength_nbytes);
/* HitReplay input (replaces hephaistos IOQueue(mode="update") of response.py:357-359): copies a
 * host queue (uint32 count + SoA columns of `capacity` entries, layout of thx_hit_queue_bytes
 * (capacity, polarized)) to the device queue the next THX_TRACER_HIT_REPLAY runs read */
int thx_tracer_write_hits(thx_tracer* tracer, const void* src, size_t nbytes);
/* parity probe: per-event dump of the last run, [batch][path_length+3][THX_TRACK_WORDS] */
int thx_tracer_read_track(thx_tracer* tracer, float* out, size_t n_floats);
/* launch geometry chosen by the library: workgroups, threads per workgroup,
 * dynamic LDS bytes, registers are reported by rocprof */
int thx_tracer_launch_info(thx_tracer* tracer, uint32_t batch, uint32_t* grid, uint32_t* block,
                           uint32_t* lds_bytes);
/* timing of the most recent thx_tracer_run on cfg, measured with HIP events on
 * the launch stream (ms); synchronises */
int thx_tracer_last_kernel_ms(thx_tracer* tracer, uint32_t cfg, float* ms);
/* durations (ms) of the last min(n, 64) tracing-kernel launches, oldest first; every
 * launch is bracketed by a HIP event pair on its launch stream; synchronises */
int thx_tracer_kernel_times(thx_tracer* tracer, float* ms_out, uint32_t n, uint32_t* n_written);

/* ---- RNG probe (replaces RNGBufferSink, random.py:44-226): out[stream][sample]
 * = raw uint32 behind random(base_stream+stream, first_dim+sample) ---- */
int thx_rng_fill(thx_ctx* ctx, const thx_philox_params* p, uint32_t base_stream,
                 uint32_t n_streams, uint32_t first_dim, uint32_t n_samples, uint32_t* out_u32,
                 float* out_f32);

/* ---- target / guide probes (replace theia.testing.TargetSampler / TargetGuideSampler,
 * src/theia/testing.py:439-640; shader/target.sample.glsl, target_guide.sample.glsl): stream i draws
 * an observer in the box and a direction on the unit sphere, then samples and intersects the target
 * (`flags`: THX_FLAG_TARGET_* / THX_FLAG_GUIDE_*). target_out[i][25] = observer3 direction3
 * samplePos3 sampleNrm3 sampleProb sampleValid(u32) sampleError(i32) hitPos3 hitNrm3 hitProb
 * hitValid(u32) hitError(i32) occluded(u32); guide_out[i][13] = observer3 sampleDir3 sampleDist
 * sampleProb evalDir3 evalDist evalProb ---- */
int thx_probe_target(thx_ctx* ctx, const thx_philox_params* philox, uint32_t flags, const thx_target_params* target,
                     const float box_min[3], const float box_max[3], uint32_t n, float* target_out);
int thx_probe_guide(thx_ctx* ctx, const thx_philox_params* philox, uint32_t flags, const thx_guide_params* guide,
                    const float box_min[3], const float box_max[3], uint32_t n, float* guide_out);

/* ---- unit probe (replaces the reference's shader unit-test kernels, tests/shader/<name>.test.glsl driven by
 * tests/test_lookup.py:41-65, test_scatter.py:15-331, test_polarization.py:14-255, test_math.py:8-38): applies ONE
 * device function of the path to n inputs. in[n][n_in] f32 -> out[n][n_out] f32; `addr` = device address of the
 * table (LOOKUP) or medium (LOOKUP_MEDIUM, SCATTER_*, PHASE_MATRIX), else 0. Columns per function:
 *   LOOKUP         in u, nullValue                      out value
 *   LOOKUP_MEDIUM  in lambda                            out n, vg, mu_s, mu_e
 *   SCATTER_DIR    in dir3, cos_theta, phi              out dir3
 *   FRESNEL        in cos_i (signed), n_i, n_t          out r_s, r_p
 *   SCATTER_PROB   in inDir3, outDir3                   out p
 *   SCATTER_SAMPLE in inDir3, u0, u1                    out dir3, p
 *   ROTATE_POLREF  in dir3, ref3, newNormal3            out ref3, cos 2phi, sin 2phi
 *   ALIGN_POLREF   in dir3, oldRef3, newRef3            out cos 2phi, sin 2phi
 *   PHASE_MATRIX   in cos_theta, cos 2phi, sin 2phi, stokes4   out stokes4 = phase * rotate * stokes
 *   POLARIZER      in p, s, stokes4                     out stokes4
 *   SIGN_BIT       in x                                 out +-1 */
enum { THX_UNIT_LOOKUP = 0, THX_UNIT_LOOKUP_MEDIUM, THX_UNIT_SCATTER_DIR, THX_UNIT_FRESNEL, THX_UNIT_SCATTER_PROB,
       THX_UNIT_SCATTER_SAMPLE, THX_UNIT_ROTATE_POLREF, THX_UNIT_ALIGN_POLREF, THX_UNIT_PHASE_MATRIX, THX_UNIT_POLARIZER,
       THX_UNIT_SIGN_BIT,
       /* the Cherenkov family (tests/test_light.py:725-836 drive the first two through their own test shaders):
        * (n, a, b, u) -> cos_theta;  (n, a, b, u, cos_min, cos_max) -> (cos_theta, contrib);  (n, a, b, cos_theta) -> pdf;
        * (n, lambda) -> Frank-Tamm in eV, in photons (both per azimuth), in photons over all azimuths */
       THX_UNIT_EMISSION_ANGLE, THX_UNIT_EMISSION_ANGLE_RANGE, THX_UNIT_EMISSION_PDF, THX_UNIT_FRANK_TAMM,
       THX_UNIT_COUNT };
int thx_probe_unit(thx_ctx* ctx, uint32_t which, uint64_t addr, const float* in, uint32_t n_in, uint32_t n, float* out, uint32_t n_out);

/* RNGBufferSink over a SobolQRNG (random.sink.glsl with random.sobol.glsl): out[stream][sample] =
 * random_s(base_stream + stream, first_dim + sample); sample_dim 2: out[stream][sample][2] = random2D_s */
int thx_rng_fill_sobol(thx_ctx* ctx, uint32_t seed, uint32_t base_stream, uint32_t n_streams, uint32_t first_dim,
                       uint32_t n_samples, uint32_t sample_dim, float* out_f32);

/* ---- multi-GPU exchange step (the reference is single-device, nothing to replace):
 * photons shard by index range (thx_run_params.photon_offset), the per-GPU time histograms are summed with
 * ONE RCCL all-reduce. RCCL is loaded on first use. Rank 0 makes the id (thx_comm_unique_id) and hands the
 * 128 bytes to the other ranks by any host channel (file, socket, torch.distributed store); every rank then
 * calls thx_comm_init with its own context. thx_hist_allreduce sums n_floats f32 at device address d_hist
 * in place over all ranks, on `stream` (NULL = the context's stream); it is asynchronous like a launch. */
#define THX_COMM_ID_BYTES 128
int thx_comm_unique_id(uint8_t out[THX_COMM_ID_BYTES]);
int thx_comm_init(thx_ctx* ctx, const uint8_t unique_id[THX_COMM_ID_BYTES], int rank, int world_size, thx_comm** out);
int thx_comm_destroy(thx_comm* comm);
int thx_hist_allreduce(thx_comm* comm, uint64_t d_hist, size_t n_floats, void* stream);
/* SURVEY.md 8(e), the other two exchanges it names. thx_stats_allreduce: sum of n u64 counters at d_counters over all
 * ranks, in place (the 12 event counters of callback.stat.glsl:7-20 + the traversal counters: thx_tracer_stats_device_ptr).
 * thx_hist_gather_reduce: the deterministic form of the histogram sum -- an all-gather of the ranks' curves in rank order
 * followed by the slab reducer's fixed-order Kahan sum (estimator.reduce.glsl:17-35) over them: every rank ends with the
 * same bits in d_hist, run after run, whatever order RCCL's ring would have added them in. Asynchronous on `stream`. */
int thx_stats_allreduce(thx_comm* comm, uint64_t d_counters, size_t n, void* stream);
int thx_hist_gather_reduce(thx_comm* comm, uint64_t d_hist, size_t n_floats, void* stream);

#ifdef __cplusplus
}
#endif
#endif
