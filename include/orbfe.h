/* orbfe.h -- C ABI of the MI355X-native ORB feature front end (liborbfe.so).
 *
 * Drop-in boundary for the per-frame hot path of AlejandroSilvestri/os1 (ORB-SLAM2 fork):
 * ORBextractor::operator() and the windowed Hamming searches of ORBmatcher.  Each entry point
 * names the reference interface it replaces (paths relative to the reference root).  The C++
 * host side in include/orbfe/ORBextractor.h and include/orbfe/orb_shim.hpp reproduces the ORB_SLAM2:: interfaces
 * on top of this ABI; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C, POD structs, caller-allocated outputs, no exceptions;
 *   - every function returns ORBFE_OK (0) or a negative error code; orbfe_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread;
 *   - a handle is single-threaded (like the reference's ORBextractor instance, which mutates
 *     mvImagePyramid and is only ever called from the Tracking thread); use one handle per
 *     GPU / host thread;
 *   - there is NO CPU fallback: if no gfx950 device is usable, create() fails with
 *     ORBFE_ERR_NO_DEVICE.
 */
#ifndef ORBFE_H_
#define ORBFE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBFE_OK 0
#define ORBFE_ERR_INVALID (-1)   /* bad argument (null pointer, non-positive size, cap too small) */
#define ORBFE_ERR_NO_DEVICE (-2) /* no usable HIP device / wrong architecture                      */
#define ORBFE_ERR_HIP (-3)       /* a HIP runtime call failed; see orbfe_last_error()              */
#define ORBFE_ERR_TOO_SMALL (-4) /* image too small: some pyramid level has <1 FAST cell column/row
                                    (the reference divides by zero there, ORBextractor.cc:820-823) */
#define ORBFE_ERR_OVERFLOW (-5)  /* an output capacity was exceeded (outputs truncated)            */

/* Same 28-byte layout as cv::KeyPoint: pt.x, pt.y, size, angle, response, octave, class_id. */
typedef struct OrbfeKeyPoint {
  float x, y;      /* pt, in level-0 pixel units (level coords * scale, ORBextractor.cc:959-965) */
  float size;      /* (float)(int)(31 * scale[octave])                    (ORBextractor.cc:879,888) */
  float angle;     /* IC-angle in degrees [0,360)                         (ORBextractor.cc:86-113)  */
  float response;  /* FAST score                                                                    */
  int32_t octave;  /* pyramid level                                                                 */
  int32_t class_id; /* always -1                                                                    */
} OrbfeKeyPoint;

typedef struct orbfe_extractor orbfe_extractor;
typedef struct orbfe_matcher orbfe_matcher;
typedef struct orbfe_vocabulary orbfe_vocabulary;   /* bag-of-words section below */

const char* orbfe_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unusable). Does not create a context. */
int orbfe_device_count(void);

/* Device-memory helpers so a caller without its own HIP code can keep frames resident in HBM
 * (orbfe_extract_batch with in_device_memory != 0).  Plain hipMalloc/hipFree/hipMemcpy/
 * hipDeviceSynchronize on the given device. */
int orbfe_device_malloc(int device_id, size_t bytes, void** out);
int orbfe_device_free(int device_id, void* ptr);
int orbfe_device_upload(int device_id, void* dst_device, const void* src_host, size_t bytes);
int orbfe_device_synchronize(int device_id);
/* Page-locked host memory for frames handed over in HOST memory (in_device_memory == 0): a frame that
 * lives in such a buffer is copied by DMA straight from it while earlier batches compute; a frame in
 * ordinary pageable memory is staged by the HIP runtime first (blocking the caller; 18 k vs 25 k frames/s at
 * 1080p end to end).  hipHostMalloc /
 * hipHostFree.  A cv::Mat can wrap the buffer (Mat(rows, cols, CV_8UC1, ptr, step)) so the camera/decoder
 * writes into it directly. */
int orbfe_host_alloc(size_t bytes, void** out);
int orbfe_host_free(void* ptr);
/* The same for a buffer the caller already owns (a capture ring, a decoder's output, the data of a long-lived cv::Mat): page-locks
 * [ptr, ptr + bytes) and maps it for the GPU (hipHostRegister); frames inside it then take the page-locked route -- for the one-frame
 * call of Frame.cc:133 that is 0.15 ms instead of 0.19 ms at 1080p.  Registering costs hundreds of microseconds: do it once per
 * buffer, not per frame, and unregister BEFORE the memory is freed (a freed-and-reallocated range that is still registered would be
 * read through its old pages). */
int orbfe_host_register(void* ptr, size_t bytes);
int orbfe_host_unregister(void* ptr);
/* NUMA placement for multi-GPU hosts (one process or thread group per GPU): the NUMA node the
 * device hangs off (sysfs numa_node of its PCI function; -1 = unknown), and a helper that restricts the CALLING
 * thread to that node's CPUs (intersected with its current mask).  Threads created and page-locked buffers first
 * touched afterwards -- the stream runner's workers, orbfe_host_alloc memory -- then live next to the GPU.
 * Returns the number of CPUs in the new mask, 0 if nothing was changed (node unknown). */
int orbfe_device_numa_node(int device_id);
int orbfe_bind_thread_to_device(int device_id);
/* Host-to-device link rate seen by `reps` back-to-back copies of `bytes` from `host` (page-locked for a DMA figure),
 * in GB/s: the bound of any caller that hands over host frames (bench.py's pcie_inclusive leg). */
int orbfe_debug_h2d_rate(int device_id, const void* host, size_t bytes, int reps, double* gb_per_s);

/* ---------------------------------------------------------------------------------------------
 * Extractor.  Replaces ORB_SLAM2::ORBextractor (include/ORBextractor.h:155-373).
 * ------------------------------------------------------------------------------------------- */

/* ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
 * int minThFAST)  (include/ORBextractor.h:164-165, src/ORBextractor.cc:442-502).
 * device_id: HIP device ordinal the handle is bound to. */
int orbfe_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                           int device_id, orbfe_extractor** out);
void orbfe_extractor_destroy(orbfe_extractor* h);

/* GetLevels / GetScaleFactor (include/ORBextractor.h:194-207). */
int orbfe_extractor_levels(const orbfe_extractor* h);
int orbfe_extractor_device(const orbfe_extractor* h);   /* the device_id given at creation; -1 for NULL */
float orbfe_extractor_scale_factor(const orbfe_extractor* h);
/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (include/ORBextractor.h:213-239).  Each output has nlevels floats; any may be NULL. */
int orbfe_extractor_scale_tables(const orbfe_extractor* h, float* scale, float* inv_scale, float* sigma2,
                                 float* inv_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:467-478); nlevels ints. */
int orbfe_extractor_features_per_level(const orbfe_extractor* h, int32_t* out);
/* Upper bound on keypoints one frame can return: sum over levels of max(N_l + 2, 16) -- a level overshoots its quota
 * N_l by at most 2, but DistributeOctTree divides every root node once before it looks at N_l (src/ORBextractor.cc:
 * 620-700), so a level with a tiny quota still returns up to 4 * roots keypoints (roots = round(width/height) <= 4
 * for aspect ratios below 4.5; wider strips are reported with ORBFE_ERR_OVERFLOW if they exceed `cap`).  Equals
 * nfeatures + 2*nlevels whenever every N_l >= 14.  Size `cap` with it. */
int orbfe_extractor_max_keypoints(const orbfe_extractor* h);
/* The same bound for a given image size (exact number of quadtree roots per level): needed only for strips wider
 * than 4.5 : 1 or to size buffers tightly. */
int orbfe_extractor_max_keypoints_for_size(const orbfe_extractor* h, int rows, int cols);

/* How orbfe_extract_batch_collect* (and the blocking calls built on it) wait for the GPU.  poll_us = 0 (default): the
 * runtime's own wait, which spins -- lowest latency, one busy host core per waiting thread.  poll_us > 0: poll the
 * stream and sleep poll_us microseconds between polls; the result arrives up to poll_us later and the core is free in
 * between.  The stream runner (orbfe_stream_*) sets 50 on its handles (ORBFE_POLL_WAIT_US overrides): with several
 * batches in flight the delay is hidden and a rank of a multi-GPU run needs 1.5 instead of 2 host cores.  Not allowed
 * between a submit and its collect. */
int orbfe_extractor_set_wait_mode(orbfe_extractor* h, int poll_us);

/* Colour input (Tracking::GrabImageMonocular, src/Tracking.cc:96-109: cvtColor(RGB2GRAY / BGR2GRAY) on 3- and
 * 4-channel images before the Frame is built).  After this call `gray` in the extract calls points to interleaved
 * 8-bit pixels of the given format (stride in bytes, cols in pixels); the conversion runs on the GPU in front of the
 * pyramid: gray = (R*cr + G*cg + B*cb + half) >> shift, alpha ignored -- OpenCV's 8-bit RGB2Gray with 15-bit
 * coefficients {9798, 19235, 3735} (ORBFE_GRAY_Q15, OpenCV >= 4.1.1) or 14-bit {4899, 9617, 1868}
 * (ORBFE_GRAY_Q14, earlier releases). */
enum { ORBFE_INPUT_GRAY8 = 0, ORBFE_INPUT_RGB8 = 1, ORBFE_INPUT_BGR8 = 2, ORBFE_INPUT_RGBA8 = 3, ORBFE_INPUT_BGRA8 = 4 };
enum { ORBFE_GRAY_Q15 = 0, ORBFE_GRAY_Q14 = 1 };
int orbfe_extractor_set_input_format(orbfe_extractor* h, int format, int gray_variant);

/* Which cv::GaussianBlur(7x7, sigma 2) the descriptors are sampled from (src/ORBextractor.cc:949-950 -- it decides every descriptor
 * bit).  OpenCV's 8-bit path is 8.8 fixed point; the taps depend on the release the reference was BUILT with (nothing in the
 * reference pins it: .cproject:40 names "opencv4", README.md:18 dates the project Feb 2019 = the 4.0.x era):
 *   ORBFE_GAUSS_ED       [18,34,48,56,48,34,18] / 256, rounding error diffused so that the taps add up to 256 -- OpenCV >= 4.1.1
 *                        (and 3.4.7+); the default;
 *   ORBFE_GAUSS_ROUNDED  [18,34,49,55,49,34,18] (sum 257), every tap rounded on its own, and the saturating final cast that
 *                        arithmetic needs: min(255, (sum + 32768) >> 16) -- OpenCV 4.0.0 - 4.1.0 (and 3.4.2 - 3.4.6).
 * Both are restated from upstream from memory (parity unpinned); tests/test_opencv_live.py picks the expected one
 * from cv2.__version__ wherever an OpenCV exists.  The environment variable ORBFE_GAUSS_VARIANT=0|1 presets it for every
 * extractor created afterwards (for a caller that cannot be recompiled). */
enum { ORBFE_GAUSS_ED = 0, ORBFE_GAUSS_ROUNDED = 1 };
int orbfe_extractor_set_blur_variant(orbfe_extractor* h, int variant);

/* void ORBextractor::operator()(InputArray image, InputArray mask, vector<KeyPoint>& keypoints,
 *                               OutputArray descriptors)
 * (include/ORBextractor.h:185-187, src/ORBextractor.cc:907-969).
 *   gray/rows/cols/stride_bytes : CV_8UC1 image in HOST memory (mask is ignored by the reference);
 *   kps  [cap]                  : keypoints, level-major, quadtree list order inside a level;
 *   desc [cap*32]               : descriptors, row i belongs to kps[i] (CV_8U, 32-byte rows);
 *   *n_out                      : number of keypoints produced.
 * rows==0 || cols==0 || gray==NULL: returns ORBFE_OK with *n_out = 0 and outputs untouched
 * (the reference returns silently, ORBextractor.cc:910-911). */
int orbfe_extract(orbfe_extractor* h, const uint8_t* gray, int rows, int cols, size_t stride_bytes,
                  OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out);

/* Batched form of the same call for independent frames of one size (throughput path: one
 * kernel launch sequence for all frames).  gray[i] points to frame i; in_device_memory != 0
 * means the frame pointers are DEVICE pointers on the handle's GPU (frames already resident in
 * HBM).  kps: nframes*cap entries, desc: nframes*cap*32 bytes, n_out: nframes ints. */
int orbfe_extract_batch(orbfe_extractor* h, int nframes, const uint8_t* const* gray, int in_device_memory,
                        int rows, int cols, size_t stride_bytes, OrbfeKeyPoint* kps, uint8_t* desc, int cap,
                        int* n_out);

/* Asynchronous form of orbfe_extract_batch: _submit enqueues the whole extractor for the batch on the
 * handle's stream and returns at once; _collect waits for it and fills the outputs exactly as
 * orbfe_extract_batch does.  One batch may be outstanding per handle; the frames (and, for host input,
 * the host buffers) must stay valid until _collect returns.  Lets a caller overlap the GPU extraction of
 * batch k+1 with the matching of batch k. */
int orbfe_extract_batch_submit(orbfe_extractor* h, int nframes, const uint8_t* const* gray, int in_device_memory,
                               int rows, int cols, size_t stride_bytes);
int orbfe_extract_batch_collect(orbfe_extractor* h, OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out);
/* Waits until the GPU has finished the submitted batch WITHOUT collecting it: a pipelined caller submits its next batch on another
 * handle between this call and the collect, so that the host-side assembly of the results (0.25 ms for 64 1080p frames) does not
 * leave a gap in the GPU's queue (the stream runner does; orbfe_stream_*).  The collect call that follows returns without waiting. */
int orbfe_extract_batch_wait(orbfe_extractor* h);

/* Extraction + GPU-resident matching for consecutive frames of ONE stream.  A chain carries the predecessor
 * (the previous batch's last frame, kept in HBM) across batches; the batches of a stream may alternate between
 * extractor handles of identical configuration on the same device, but must be submitted in stream order.
 * _submit_matched = orbfe_extract_batch_submit followed, on the GPU, by
 *   ORBmatcher(nnratio, check_orientation).SearchForInitialization(F1 = predecessor, F2 = frame,
 *       vbPrevMatched := F1's keypoints (Tracking.cc:355-357), vnMatches12, window_size)   (ORBmatcher.cc:400-515)
 * for every frame of the batch, reading keypoints, angles and descriptors straight from the extractor's result
 * arena.  _collect_matched additionally returns matches12 [nframes][cap] (row f = vnMatches12 of frame f, indexed
 * by the predecessor's keypoints, -1 = none) and nmatches [nframes] (0 for a stream's very first frame). */
typedef struct orbfe_sfi_chain orbfe_sfi_chain;
int orbfe_sfi_chain_create(const orbfe_extractor* h, orbfe_sfi_chain** out);
void orbfe_sfi_chain_destroy(orbfe_sfi_chain* c);
/* isolated != 0: every batch submitted with this chain stands alone -- its first frame has no predecessor (its match vector is all -1,
 * its count 0) and its last frame is not handed on.  For callers that deal the batches of ONE stream to several devices and match the
 * boundary pairs themselves (orbfe_stream_multi_*). */
int orbfe_sfi_chain_set_isolated(orbfe_sfi_chain* c, int isolated);
int orbfe_extract_batch_submit_matched(orbfe_extractor* h, orbfe_sfi_chain* chain, int nframes, const uint8_t* const* gray,
                                       int in_device_memory, int rows, int cols, size_t stride_bytes, const float bounds[4],
                                       int window_size, float nnratio, int check_orientation);
int orbfe_extract_batch_collect_matched(orbfe_extractor* h, OrbfeKeyPoint* kps, uint8_t* desc, int cap, int* n_out,
                                        int32_t* matches12, int* nmatches);

/* Stage accessors for the parity tests (state of the LAST extract call, frame index in batch). */
int orbfe_debug_level_size(const orbfe_extractor* h, int level, int* w, int* hgt);
int orbfe_debug_level_copy(orbfe_extractor* h, int frame, int level, uint8_t* out /* w*h, tight */);
/* FAST candidates of one level BEFORE the quadtree, reference order (cell-row-major, row-major in
 * a cell): triples (x, y, score) with x,y relative to (minBorderX,minBorderY) = (16,16) as in
 * vToDistributeKeys (ORBextractor.cc:861-866). */
int orbfe_debug_candidates(orbfe_extractor* h, int frame, int level, int32_t* xys, int cap, int* n_out);
/* Host wall-clock milliseconds of the last call: [0]=waiting for stage 1 (pyramid+FAST+compaction) and
 * the candidate D2H, [1]=host quadtrees + selection packing, [2]=tail wait for stage 2
 * (orientation+blur+rBRIEF), [3]=output assembly, [4]=total. */
int orbfe_debug_stage_ms(const orbfe_extractor* h, float out[5]);
/* GPU time (HIP events recorded on the launch stream) accumulated per kernel group since the last
 * reset: out_ms[0]=pyramid (k_resize x (nlevels-1)), [1]=k_fast_tasks, [2]=k_compact,
 * [3]=k_describe, [4]=k_quadtree; *batches = launches of each group, *frames = frames processed. */
int orbfe_debug_kernel_ms(orbfe_extractor* h, double out_ms[5], long long* batches, long long* frames, int reset);
/* k_fast_tasks (the dominant kernel) is timed in every call of MORE than ORBFE_CONE_MAX_FRAMES (default 2) frames;
 * calls with one or two frames take the latency route, which records no events (each costs a dependent-launch gap),
 * so orbfe_debug_kernel_ms reports frames = 0 for them.  enable != 0 also times the other groups, at the price of an
 * event (a few microseconds of stream gap) between them. */
int orbfe_debug_set_profiling(orbfe_extractor* h, int enable);
/* Measurement only: needs a library built with `make EXPERIMENTS=1` (-DORBFE_EXPERIMENTS) and ORBFE_FAST_ABLATE=4 in the environment,
 * which selects an instantiation of the FAST kernel with s_memtime stamps between its phases (a default build returns ORBFE_ERR_INVALID): shader
 * cycles summed over every wave since the last reset -- [0] entry -> geometry known, [1] -> ROI in LDS, [2] -> pre-test done,
 * [3] -> scores done, [4] -> end, [5] = waves counted.  tools/fast_phases.py prints the per-wave averages. */
int orbfe_debug_fast_stamps(orbfe_extractor* h, unsigned long long out[8], int reset);
/* Measurement only (a `make EXPERIMENTS=1` build with ORBFE_SFI_DEBUG=1 in the environment when the process starts; a default build
 * reports no records): one record of 8 ints per pair the GPU-resident
 * SearchForInitialization resolved since the last reset -- frame in its batch, rounds of the fixed point, candidate entries, 1 if
 * the serial finish ran, shader cycles of the block, n1, n2, 1 if the candidate pool fitted LDS (tools/sfi_rounds.py). */
int orbfe_debug_sfi_records(orbfe_extractor* h, int32_t* out, int cap_records, int* n_out, int reset);
/* Device-side restatement of (cosf, sinf)(angle_deg * pi/180) used by the rBRIEF kernel, evaluated
 * on the GPU for n angles (parity test against host libm). */
int orbfe_debug_sincos(orbfe_extractor* h, const float* angle_deg, int n, float* cos_out, float* sin_out);

/* Host-only test hooks (no GPU needed): the product's own quadtree (DistributeOctTree,
 * src/ORBextractor.cc:571-795) on candidate triples (x, y relative to (minX,minY); score), returning
 * the indices of the retained candidates in list order; and the (cosf,sinf) restatement compiled
 * for the host, compared against libm over the float bit patterns [lo_bits, hi_bits] step `step`
 * (returns the number of mismatching values in *mismatches). */
int orbfe_debug_quadtree(const int16_t* x, const int16_t* y, const uint8_t* score, int n, int min_x, int max_x,
                         int min_y, int max_y, int n_target, int32_t* out_idx, int cap, int* n_out);
int orbfe_debug_sincos_host_check(uint32_t lo_bits, uint32_t hi_bits, uint32_t step, long long* mismatches);

/* ---------------------------------------------------------------------------------------------
 * Stream runner (throughput path; no counterpart class in the reference, whose Tracking thread drives
 * ORBextractor / ORBmatcher one frame at a time -- System.cc:115-152, Tracking.cc:92-121,344-419).
 * A stream owns `depth` extractor handles + one matcher on one GPU and two native worker threads:
 * batches pushed with orbfe_stream_push are extracted (asynchronous submit/collect, `depth` batches in
 * flight) and every frame is matched against its predecessor in the stream with
 * SearchForInitialization (vbPrevMatched := predecessor keypoints, as Tracking.cc:355-357); results come
 * back in push order from orbfe_stream_pop.  Identical results to calling orbfe_extract_batch and
 * orbfe_search_for_initialization frame by frame.
 * ------------------------------------------------------------------------------------------- */
typedef struct orbfe_stream orbfe_stream;
/* batch = frames per push; depth = extraction batches in flight (1..8).
 * HARDWARE QUEUES: each batch in flight has its own HIP stream and the HIP runtime folds a process's streams onto GPU_MAX_HW_QUEUES
 * hardware queues (4 unless the ENVIRONMENT variable said otherwise before the process's first HIP call: the runtime reads it once,
 * a library cannot set it for its host process; bench.py sets 8 before importing torch).  Two streams that share a queue serialise,
 * so the runner measures at creation how many of its streams run side by side (about a millisecond) and keeps that many batches
 * in flight: 3 in a process with the default 4 queues (0.97 of the rate of depth 4 on 8 queues), `depth` where the queues are there.
 * orbfe_stream_batches_in_flight reports the number; a line on stderr says so once (ORBFE_QUIET=1 silences it). */
int orbfe_stream_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device_id,
                        int batch, int depth, orbfe_stream** out);
/* Batches the GPU is working on are waited for; batches still queued are dropped without being submitted.  Frames of
 * submitted batches must stay valid until destroy returns (or pop every batch first). */
void orbfe_stream_destroy(orbfe_stream* s);
/* bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}; window_size <= 0 disables matching (extract only).  Only while no
 * batch is in flight (every pushed batch popped); each batch carries the parameters in force at its push. */
int orbfe_stream_set_matching(orbfe_stream* s, const float bounds[4], int window_size, float nnratio,
                              int check_orientation);
/* isolated != 0: frame 0 of every pushed batch has no predecessor (orbfe_sfi_chain_set_isolated on the runner's chain); only while
 * no batch is in flight. */
int orbfe_stream_set_isolated_batches(orbfe_stream* s, int isolated);
int orbfe_stream_batches_in_flight(const orbfe_stream* s);
/* orbfe_stream_pop without the "valid until the next pop" rule: the result stays valid until orbfe_stream_release(s, *ticket).  Any
 * number of results may be held; each keeps one of the runner's result slots (orbfe_stream_queue_slots) busy, and a push waits while no
 * slot is free.  Do not mix with orbfe_stream_pop on the same runner. */
int orbfe_stream_pop_hold(orbfe_stream* s, const OrbfeKeyPoint** kps, const uint8_t** desc, const int** n_kps, const int32_t** matches12,
                          const int** nmatches, int* ticket);
int orbfe_stream_release(orbfe_stream* s, int ticket);
/* orbfe_extractor_set_input_format for every extractor of the runner (only while no batch is in flight). */
int orbfe_stream_set_input_format(orbfe_stream* s, int format, int gray_variant);
/* orbfe_extractor_set_blur_variant for every extractor of the runner (only while no batch is in flight). */
int orbfe_stream_set_blur_variant(orbfe_stream* s, int variant);
/* orbfe_extractor_set_vocabulary for every extractor of the runner (only while no batch is in flight); afterwards
 * orbfe_stream_bow_raw returns, for frame `frame` of the LAST POPPED batch, the per-keypoint (leaf node, level node)
 * pairs (pointers into the runner's buffers, valid until the next pop; feed them to orbfe_bow_assemble). */
int orbfe_stream_set_vocabulary(orbfe_stream* s, orbfe_vocabulary* v, int levelsup);
int orbfe_stream_bow_raw(orbfe_stream* s, int frame, const uint32_t** leaf_node, const uint32_t** level_node, int* n);
/* Per-frame output capacity (keypoints) of the arrays returned by orbfe_stream_pop = their row stride.  It grows when a
 * push brings frames of a geometry that can return more keypoints (orbfe_extractor_max_keypoints_for_size); such a push
 * is only accepted while no batch is in flight. */
int orbfe_stream_capacity(const orbfe_stream* s);
/* Result slots of the runner = batches that can be pushed ahead of the pops before orbfe_stream_push blocks, plus 2
 * (one is in the caller's hands after a pop, one is being filled).  Default depth+4; a caller whose own thread may be
 * held up for milliseconds asks for more (only while no batch is in flight; the number never shrinks;
 * depth+2 <= nslots <= 256).  A caller must never push more than orbfe_stream_queue_slots()-2 batches ahead of its
 * pops from the popping thread: no slot could become free and the push would wait forever. */
int orbfe_stream_set_queue_slots(orbfe_stream* s, int nslots);
int orbfe_stream_queue_slots(const orbfe_stream* s);
/* Enqueue one batch (`batch` frame pointers; device pointers if in_device_memory != 0).  Returns at once unless every
 * result slot is taken (see orbfe_stream_set_queue_slots).  The frames must stay valid until their batch is popped. */
int orbfe_stream_push(orbfe_stream* s, const uint8_t* const* gray, int in_device_memory, int rows, int cols,
                      size_t stride_bytes);
/* Wait for the oldest batch; ORBFE_ERR_INVALID if no pushed batch is outstanding.  Output pointers stay valid until
 * the next pop:
 *   kps [batch][cap], desc [batch][cap][32], n_kps [batch];
 *   matches12 [batch][cap]: row i = vnMatches12 of (F1 = frame i-1 of the stream, F2 = frame i), indexed by
 *   F1's keypoints (first n_kps[i-1] entries valid; row 0 uses the last frame of the previous batch);
 *   nmatches [batch]: return values of the searches (0 for the very first frame of the stream). */
int orbfe_stream_pop(orbfe_stream* s, const OrbfeKeyPoint** kps, const uint8_t** desc, const int** n_kps,
                     const int32_t** matches12, const int** nmatches);
/* Host wall-clock ms the workers spent in submit / collect (incl. waiting for the GPU) / match calls, and the
 * number of finished batches, since the last reset: out = {submit, collect, match, batches}. */
int orbfe_stream_stats(orbfe_stream* s, double out[4], int reset);
/* Sum of orbfe_debug_kernel_ms over the stream's extractor handles. */
int orbfe_stream_kernel_ms(orbfe_stream* s, double out_ms[5], long long* batches, long long* frames, int reset);

/* ---------------------------------------------------------------------------------------------
 * Matcher.  Replaces the hot subset of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:71-270).
 * ------------------------------------------------------------------------------------------- */

/* static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)
 * (include/ORBmatcher.h:74, src/ORBmatcher.cc:1605-1621): Hamming distance of two 256-bit rows.
 * Scalar host helper (callers such as MapPoint.cc:266 use it on single pairs). */
int orbfe_hamming(const uint8_t a[32], const uint8_t b[32]);

int orbfe_matcher_create(int device_id, orbfe_matcher** out);
void orbfe_matcher_destroy(orbfe_matcher* m);
int orbfe_matcher_device(const orbfe_matcher* m);       /* the device_id given at creation; -1 for NULL */
/* For callers that keep query-side tables in device memory across searches (the local map's MapPoint descriptors,
 * src/Tracking.cc:818-824 sends the same few thousand every frame; include/orbfe/orb_shim.hpp's MatcherContext does):
 * a host-to-device copy enqueued on the matcher's own stream, i.e. ordered BEFORE the matcher's next search and after its
 * previous one.  Returns at once; `src_host` must be page-locked (orbfe_host_alloc) and unchanged until
 * orbfe_matcher_synchronize(m) or the next search through m has returned. */
int orbfe_matcher_upload_async(orbfe_matcher* m, void* dst_device, const void* src_host, size_t bytes);
int orbfe_matcher_synchronize(orbfe_matcher* m);

/* int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched,
 *                                         vector<int>& vnMatches12, int windowSize)
 * (include/ORBmatcher.h:193, src/ORBmatcher.cc:400-515) with ORBmatcher(nnratio, checkOri).
 *   kps1/desc1/n1   : F1.mvKeysUn, F1.mDescriptors
 *   kps2/desc2/n2   : F2.mvKeysUn, F2.mDescriptors
 *   bounds          : {Frame::mnMinX, mnMaxX, mnMinY, mnMaxY} (static image bounds, Frame.cc:322-353)
 *   prev_xy [2*n1]  : vbPrevMatched, in/out
 *   matches12 [n1]  : vnMatches12, out (-1 = none)
 *   *nmatches       : return value of the reference function */
int orbfe_search_for_initialization(orbfe_matcher* m, const OrbfeKeyPoint* kps1, const uint8_t* desc1, int n1,
                                    const OrbfeKeyPoint* kps2, const uint8_t* desc2, int n2,
                                    const float bounds[4], float* prev_xy, int32_t* matches12, int window_size,
                                    float nnratio, int check_orientation, int* nmatches);

/* Batched form for throughput callers: npairs independent (F1, F2) pairs, one upload + one kernel launch +
 * one download for all of them, bookkeeping of the pairs resolved in parallel on the host.  Arrays of
 * per-pair pointers/sizes; bounds are shared (same camera). nmatches: npairs ints. */
int orbfe_search_for_initialization_batch(orbfe_matcher* m, int npairs, const OrbfeKeyPoint* const* kps1,
                                          const uint8_t* const* desc1, const int* n1,
                                          const OrbfeKeyPoint* const* kps2, const uint8_t* const* desc2,
                                          const int* n2, const float bounds[4], float* const* prev_xy,
                                          int32_t* const* matches12, int window_size, float nnratio,
                                          int check_orientation, int* nmatches);

/* MapPoint flag bits for the searches below. */
#define ORBFE_MP_IN_VIEW 1u    /* MapPoint::mbTrackInView                                   */
#define ORBFE_MP_BAD 2u        /* MapPoint::isBad()                                         */
#define ORBFE_MP_CANDIDATO 4u  /* MapPoint::plCandidato (os1 far-point extension)           */
#define ORBFE_MP_OBSERVED 8u   /* MapPoint::Observations() > 0                              */

/* int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)
 * (include/ORBmatcher.h:93, src/ORBmatcher.cc:45-132) with ORBmatcher(nnratio).
 * The facade snapshots the MapPoint fields (under the reference's accessors) into flat arrays:
 *   mp_proj_xy [2*n_mp] : mTrackProjX, mTrackProjY       mp_level [n_mp] : mnTrackScaleLevel
 *   mp_viewcos [n_mp]   : mTrackViewCos                  mp_flags [n_mp] : ORBFE_MP_* bits
 *   mp_desc [32*n_mp]   : GetDescriptor()
 *   kp_occupied [n]     : 1 iff F.mvpMapPoints[i] != NULL && ->Observations() > 0 at call time
 *   kp_assigned [n] out : index into vpMapPoints written to F.mvpMapPoints[i] (last writer), -1 = untouched
 *   scale_factors       : F.mvScaleFactors (nlevels floats) */
int orbfe_search_by_projection(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n,
                               const float bounds[4], const float* scale_factors, int nlevels,
                               const uint8_t* kp_occupied, const float* mp_proj_xy, const int32_t* mp_level,
                               const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* mp_desc, int n_mp,
                               float th, float nnratio, int32_t* kp_assigned, int* nmatches);

/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th)
 * (include/ORBmatcher.h:118, src/ORBmatcher.cc:1292-1423) and
 * int ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, const float th, const int ORBdist)
 * (include/ORBmatcher.h:133, src/ORBmatcher.cc:1425-1552), from the projection onwards: the facade
 * projects each source MapPoint exactly as the reference does (float cv::Mat arithmetic,
 * ORBmatcher.cc:1326-1343 / 1451-1477) and passes
 *   src_uv [2*n_src], src_level (nLastOctave resp. nPredictedLevel), src_angle (source keypoint angle),
 *   src_flags (ORBFE_MP_OBSERVED), src_valid (0 = absent / outlier / rejected), src_desc;
 *   max_dist = TH_HIGH (100) resp. ORBdist; skip_any_occupied = 0 (skip kps whose MapPoint has
 *   observations, :1364-1366) resp. 1 (skip kps with any MapPoint, :1493-1494).
 * kp_assigned out: source index per keypoint, -1 untouched, -2 = reset to NULL by the rotation check. */
int orbfe_search_by_projection_uv(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n,
                                  const float bounds[4], const float* scale_factors, int nlevels,
                                  const uint8_t* kp_occupied, const float* src_uv, const int32_t* src_level,
                                  const float* src_angle, const uint8_t* src_flags, const uint8_t* src_valid,
                                  const uint8_t* src_desc, int n_src, float th, int max_dist,
                                  int skip_any_occupied, int check_orientation, int32_t* kp_assigned,
                                  int* nmatches);

/* Host wall-clock ms of the last search call: [0] grid sort + arena build, [1] upload + kernel + download,
 * [2] sequential bookkeeping (resolve). */
int orbfe_debug_matcher_ms(const orbfe_matcher* m, double out[3]);

/* Generic GPU primitive under every windowed search of the reference (Fuse x2
 * ORBmatcher.cc:806-1064, SearchBySim3 :1066-1290, SearchByProjection(KeyFrame*, Scw, ...) :285-398,
 * SearchForTriangulation's window variant): for nq queries (x, y, r, minLevel, maxLevel, 32-byte descriptor)
 * against one frame's keypoints, the candidate lists Frame/KeyFrame::GetFeaturesInArea would return, in
 * reference order, each with its Hamming distance to the query descriptor.  r < 0 marks an inactive query.
 *   counts[nq], offsets[nq] : candidate count and first entry of each query in `pool`
 *   pool[pool_cap]          : entries  index | distance << 16
 *   *pool_used              : entries needed; if > pool_cap the call returns ORBFE_ERR_OVERFLOW (nothing lost:
 *                             call again with a larger pool)
 * The per-function bookkeeping (chi-square gates, best/second-best, occupancy) stays with the caller, exactly
 * like orb_shim.hpp does for the searches above. */
int orbfe_window_candidates(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n,
                            const float bounds[4], int nq, const float* qx, const float* qy, const float* qr,
                            const int32_t* qmin_level, const int32_t* qmax_level, const uint8_t* qdesc,
                            uint32_t* counts, uint32_t* offsets, uint32_t* pool, size_t pool_cap, size_t* pool_used);

/* The projected best-match loop shared by int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const
 * vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th) (src/ORBmatcher.cc:285-398, loop :357-392),
 * int ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, float th) (:806-939, loop :872-936),
 * int ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, ...) (:941-1064, loop :1014-1050) and both directions of
 * int ORBmatcher::SearchBySim3(...) (:1066-1290), from KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:637-676) onwards.
 * The caller keeps the reference's projection code (cv::Mat float arithmetic, depth / viewing-angle / PredictScale
 * checks) and passes, per surviving MapPoint: src_uv, src_radius (= th * mvScaleFactors[nPredictedLevel]), src_level
 * (= nPredictedLevel; keypoints of octave [level-1, level] qualify), src_desc, src_valid (0 = rejected earlier).
 *   kp_skip (n, optional) / claim: vpMatched semantics of SearchByProjection -- a keypoint with kp_skip != 0 is never
 *     taken, and with claim != 0 an accepted keypoint is skipped by later sources (vpMatched[bestIdx] = pMP);
 *   inv_level_sigma2 (nlevels, optional) + chi2: Fuse's reprojection gate e2 * mvInvLevelSigma2[kpLevel] > 5.99;
 *   max_dist: TH_LOW (50) for SearchByProjection / Fuse, TH_HIGH (100) for SearchBySim3.
 * best_idx[i] = accepted keypoint index or -1 (the first minimum in GetFeaturesInArea order), best_dist[i] (optional)
 * its distance; *nmatches = number accepted.  The MapPoint bookkeeping (Replace / AddObservation / vnMatch agreement)
 * stays with the caller. */
int orbfe_search_projected(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                           int n_src, const float* src_uv, const float* src_radius, const int32_t* src_level,
                           const uint8_t* src_valid, const uint8_t* src_desc, const uint8_t* kp_skip, int claim,
                           const float* inv_level_sigma2, int nlevels, double chi2, int max_dist, int32_t* best_idx,
                           int32_t* best_dist, int* nmatches);

/* ---------------------------------------------------------------------------------------------
 * Device-resident frames.  The reference builds a frame's grid once (Frame::AssignFeaturesToGrid, src/Frame.cc:114-129,
 * from the constructor :111) and every search of that frame reuses it: 2-3 searches per tracked frame
 * (src/Tracking.cc:608, 614, 824), many more per keyframe.  An orbfe_frame is that object in HBM -- mvKeysUn (x, y,
 * octave, angle), mDescriptors and the cell-sorted table Frame / KeyFrame::GetFeaturesInArea walks (Frame.cc:209-262,
 * KeyFrame.cc:637-676) -- built once; the `_frame` searches below upload only their queries and return only their
 * result vector: candidate lists AND the reference's sequential bookkeeping stay on the GPU.
 * A frame belongs to one device; it may be searched through any matcher of that device, one search at a time.
 * ------------------------------------------------------------------------------------------- */
typedef struct orbfe_frame orbfe_frame;
/* From host arrays (kps_un = mvKeysUn in cv::KeyPoint layout, desc = mDescriptors rows, bounds = mnMinX, mnMaxX, mnMinY,
 * mnMaxY): one upload + the grid build, on the matcher's stream; returns without waiting. */
int orbfe_frame_create(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, const uint8_t* desc, int n, const float bounds[4],
                       orbfe_frame** out);
/* From frame `frame_index` of the extractor's LAST COLLECTED batch (orbfe_extract / _batch / _collect): keypoints and
 * descriptors are taken where the kernels left them -- nothing is uploaded but, optionally, xy_un: the undistorted
 * coordinates [2 * n] of Frame::UndistortKeyPoints (Frame.cc:286-320; NULL = mvKeysUn = mvKeys, the camera without
 * distortion, :288-292).  Keypoint order = the order the extract call returned.  Must be called before the next
 * submit / extract on that handle (its arena is reused); the frame itself then lives on independently. */
int orbfe_frame_create_from_extract(orbfe_extractor* h, int frame_index, const float bounds[4], const float* xy_un,
                                    orbfe_frame** out);
void orbfe_frame_destroy(orbfe_frame* f);
int orbfe_frame_size(const orbfe_frame* f);
int orbfe_frame_device(const orbfe_frame* f);
/* Process-wide generation number of "which host object a resident frame stands for".  A caller that caches orbfe_frame
 * handles per reference object (include/orbfe/orb_shim.hpp does, per Frame / KeyFrame) compares the generation it saw when
 * it filled its cache with the current one and drops the cache when they differ.  orbfe_resident_invalidate() starts a new
 * generation (returns it): call it where the reference recycles identities -- Tracking::Reset() sets Frame::nNextId and
 * KeyFrame::nNextId back to 0 (src/Tracking.cc:1159-1160), Osmap's map load re-creates KeyFrames with the ids stored in the
 * file and rewrites KeyFrame::nNextId (src/Osmap.cpp:586) -- from any thread; every thread's cache notices at its next
 * search.  (The shim's cache is keyed by CONTENT, so a missing call costs memory, not correctness.)  Both are lock-free. */
unsigned long long orbfe_resident_epoch(void);
unsigned long long orbfe_resident_invalidate(void);
/* The frame's descriptor rows [n][32] in DEVICE memory, keypoint order; valid while the frame lives.  Returns when the
 * rows are complete.  Usable as the query descriptor rows of a search on another frame of the same device (a caller that
 * keeps descriptors on the GPU -- its own table of MapPoint descriptors, or a frame's rows -- passes device pointers and no
 * descriptor crosses PCIe). */
const uint8_t* orbfe_frame_descriptors_device(orbfe_frame* f);
/* Test / debug: the resident content back on the host -- keypoints (x, y, angle, octave; other fields zeroed) and
 * descriptor rows in keypoint order, the keypoint indices in grid order (capacity n) and the 64*48+1 cell offsets into
 * that order (mGrid flattened: cell = ix * 48 + iy).  Any pointer may be NULL. */
int orbfe_frame_download(orbfe_frame* f, OrbfeKeyPoint* kps_un, uint8_t* desc, int32_t* grid_order, int32_t* cell_start);

/* orbfe_search_by_projection / _uv / orbfe_search_projected on a resident frame: same arguments minus the frame's
 * arrays, same results.  (The host-array forms above run through these with a transient frame owned by the matcher;
 * ORBFE_MATCH_HOST_RESOLVE=1 keeps their round-2 route -- candidate lists to the host, bookkeeping there -- for A/B
 * runs and the parity tests.)  WHERE THE INPUT ARRAYS MAY LIVE: the DESCRIPTOR ROWS may lie in ordinary host memory,
 * in page-locked host memory (orbfe_host_alloc) or in the memory of the frame's device (16-byte aligned there, e.g.
 * orbfe_frame_descriptors_device, or a table the caller maintains with orbfe_matcher_upload_async); page-locked and
 * device rows are read by the search kernel in place.  EVERY OTHER per-query array (coordinates, levels, viewing cosines /
 * radii, angles, flags) and kp_occupied / kp_skip must be HOST memory -- the library reads them too (level range, largest
 * radius) -- page-locked ones are read by the kernel in place, ordinary ones are first copied into a page-locked
 * arena (a plain memcpy each -- there is no per-query loop on the host); a device pointer for one of them is refused with
 * ORBFE_ERR_INVALID before anything is read.  ORBFE_FRAME_ZEROCOPY=0: marshal the queries on
 * the host and upload them instead.  The call returns when the kernel's last store -- the call's number, into page-locked
 * memory -- has been seen (or, failing that for 2 ms, when the stream has drained).  Limits: at most 65 535 keypoints per frame,
 * 1 048 574 queries per search, 32 pyramid levels for the in-place route (more: the queries are marshalled on the host). */
int orbfe_search_by_projection_frame(orbfe_matcher* m, orbfe_frame* f, const float* scale_factors, int nlevels,
                                     const uint8_t* kp_occupied, const float* mp_proj_xy, const int32_t* mp_level,
                                     const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* mp_desc, int n_mp,
                                     float th, float nnratio, int32_t* kp_assigned, int* nmatches);
/* orbfe_search_by_projection_frame with the MapPoints' descriptors in a TABLE THE CALLER KEEPS ON THE DEVICE across searches
 * (Tracking::SearchLocalPoints, src/Tracking.cc:818-824, sends the same few thousand MapPoints frame after frame; their
 * descriptors change only in MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:227-292).  The table has two copies of
 * 32-byte rows: desc_table_device (memory of the frame's device, orbfe_device_malloc) and desc_table_host (page-locked,
 * orbfe_host_alloc; complete and current).  MapPoint i's descriptor is row (mp_desc_row[i] & 0x7fffffff): read from the
 * device copy when bit 31 is clear, from the host copy -- over PCIe, in place -- when it is set (a row the device has not
 * received yet; the caller sends it afterwards with orbfe_matcher_upload_async).  Per MapPoint 4 bytes of index cross PCIe
 * instead of 32 of descriptor.  mp_desc_row is host memory (page-locked: read in place).  n_rows = rows both copies of the table
 * hold: an in-view, not-bad MapPoint whose row index is >= n_rows fails the call with ORBFE_ERR_INVALID (as an out-of-range level
 * does) instead of reading outside the table.  Same results as the plain form. */
int orbfe_search_by_projection_frame_rows(orbfe_matcher* m, orbfe_frame* f, const float* scale_factors, int nlevels,
                                          const uint8_t* kp_occupied, const float* mp_proj_xy, const int32_t* mp_level,
                                          const float* mp_viewcos, const uint8_t* mp_flags, const uint8_t* desc_table_device,
                                          const uint8_t* desc_table_host, const int32_t* mp_desc_row, int n_rows, int n_mp,
                                          float th, float nnratio, int32_t* kp_assigned, int* nmatches);
int orbfe_search_by_projection_uv_frame(orbfe_matcher* m, orbfe_frame* f, const float* scale_factors, int nlevels,
                                        const uint8_t* kp_occupied, const float* src_uv, const int32_t* src_level,
                                        const float* src_angle, const uint8_t* src_flags, const uint8_t* src_valid,
                                        const uint8_t* src_desc, int n_src, float th, int max_dist, int skip_any_occupied,
                                        int check_orientation, int32_t* kp_assigned, int* nmatches);
int orbfe_search_projected_frame(orbfe_matcher* m, orbfe_frame* f, int n_src, const float* src_uv, const float* src_radius,
                                 const int32_t* src_level, const uint8_t* src_valid, const uint8_t* src_desc,
                                 const uint8_t* kp_skip, int claim, const float* inv_level_sigma2, int nlevels, double chi2,
                                 int max_dist, int32_t* best_idx, int32_t* best_dist, int* nmatches);
/* Rounds the bookkeeping kernel of the last `_frame` search needed -- the most any chunk of 2 048 consecutive queries took
 * (negative: a chunk hit the bound ORBFE_RESOLVE_MAX_ROUNDS, default 48, and a serial pass on the device finished it). */
int orbfe_debug_resolve_rounds(const orbfe_matcher* m);
/* Where that kernel kept its state: 2 = tables and candidate entries in LDS, 1 = tables in LDS, 0 = global scratch (problems
 * beyond 152 KB of tables, or ORBFE_RESOLVE_GENERIC=1). */
int orbfe_debug_resolve_route(const orbfe_matcher* m);
/* Shader-clock readings (low 32 bits) at the kernel's start, after its set-up, after the fixed point, at its end. */
int orbfe_debug_resolve_phases(const orbfe_matcher* m, int out[4]);

/* void MapPoint::ComputeDistinctiveDescriptors()  (src/MapPoint.cc:227-292), from the gathered descriptor list
 * onwards, batched over MapPoints: MapPoint p owns descriptor rows [offsets[p], offsets[p+1]) of `descs`
 * (32-byte rows, the non-bad observations in std::map order); best_idx[p] = index inside its own list of the
 * descriptor with the least median distance to the rest (median = sorted[ (size_t)(0.5*(N-1)) ], first minimum
 * wins), -1 for an empty list.  All-pairs 256-bit Hamming + per-row median on the GPU, one wave per MapPoint. */
int orbfe_distinctive_descriptors(orbfe_matcher* m, int n_mp, const int32_t* offsets, const uint8_t* descs,
                                  int32_t* best_idx);

/* ---------------------------------------------------------------------------------------------
 * Bag of words.  Replaces the DBoW2 calls on the path: Frame::ComputeBoW (src/Frame.cc:277-284) ->
 * TemplatedVocabulary<FORB>::transform(features, BowVector&, FeatureVector&, levelsup)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1204, descent :1306-1347), and the two
 * ORBmatcher::SearchByBoW overloads (src/ORBmatcher.cc:154-283, 517-650).
 * ------------------------------------------------------------------------------------------- */
/* Vocabulary from the fork's binary file (loadFromBinaryFile, TemplatedVocabulary.h:1563-1640): 4 header bytes
 * {k, L, scoring, weighting} followed by 45-byte node records {int32 parent, u8 isLeaf, u8 descriptor[32],
 * f64 weight} (packed, little endian).  Node ids count from 1 in record order (0 = root), children keep record
 * order, word ids count the isLeaf records.  orbfe_vocabulary_create takes the header fields and the records
 * separately (a loader of the text / yml formats fills the same records); _from_image takes the file contents.
 * scoring: 0 L1_NORM .. 5 DOT_PRODUCT, weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (BowVector.h:36-53).
 * The tree lives in HBM (k=10, L=6: 1.1 M nodes, about 45 MB). */
int orbfe_vocabulary_create(int device_id, int k, int L, int scoring, int weighting, const void* records, int n_records,
                            orbfe_vocabulary** out);
int orbfe_vocabulary_create_from_image(int device_id, const void* image, size_t bytes, orbfe_vocabulary** out);
void orbfe_vocabulary_destroy(orbfe_vocabulary* v);
int orbfe_vocabulary_info(const orbfe_vocabulary* v, int* k, int* L, int* scoring, int* weighting, int* n_nodes,
                          int* n_words);

/* transform(features, v, fv, levelsup) for n descriptors (32-byte rows; host memory, or device memory when
 * in_device_memory != 0).  BowVector: n_words (word id ascending, value) pairs in bow_ids / bow_values (capacity
 * n).  FeatureVector: n_fv_nodes node ids ascending in fv_nodes (capacity n), fv_offsets[i]..fv_offsets[i+1]
 * (capacity n+1) delimit the node's feature indices in fv_features (capacity n), in push_back order.
 * word_of_feature / node_of_feature (optional, n each): the word id and the node id `levelsup` levels above the
 * leaf that each descriptor reached (stopped words, weight 0, are reported here but left out of both vectors).
 * Values are bit-identical to DBoW2's doubles: weights are summed in feature order, normalised in word order. */
int orbfe_bow_transform(orbfe_vocabulary* v, const uint8_t* desc, int n, int in_device_memory, int levelsup,
                        uint32_t* bow_ids, double* bow_values, int* n_words, uint32_t* fv_nodes, uint32_t* fv_offsets,
                        uint32_t* fv_features, int* n_fv_nodes, uint32_t* word_of_feature, uint32_t* node_of_feature);

/* Frame::ComputeBoW fused into the extractor: with a vocabulary set, every extract call also runs the tree descent on
 * the descriptors while they are still in HBM (right behind the descriptor kernel, same stream) and keeps the per-
 * keypoint (word, node) pairs of the last collected batch.  orbfe_extract_bow then returns frame `frame` of that batch
 * in the layout of orbfe_bow_transform -- identical results, no descriptor round trip.  v = NULL switches it off.
 * The vocabulary must outlive the extractor's use of it and live on the same device. */
int orbfe_extractor_set_vocabulary(orbfe_extractor* h, orbfe_vocabulary* v, int levelsup);
int orbfe_extract_bow(orbfe_extractor* h, int frame, uint32_t* bow_ids, double* bow_values, int* n_words, uint32_t* fv_nodes,
                      uint32_t* fv_offsets, uint32_t* fv_features, int* n_fv_nodes, uint32_t* word_of_feature,
                      uint32_t* node_of_feature);

/* The two halves of orbfe_extract_bow, for callers that keep many frames and build the vectors later (the stream runner
 * hands out the raw pairs): per keypoint the leaf node its descriptor reached and the node `levelsup` levels above it
 * (cap entries are written, *n_out = the frame's keypoint count); orbfe_bow_assemble turns such pairs into the BowVector /
 * FeatureVector (host bookkeeping only, no GPU work). */
int orbfe_extract_bow_raw(orbfe_extractor* h, int frame, uint32_t* leaf_node, uint32_t* level_node, int cap, int* n_out);
int orbfe_bow_assemble(orbfe_vocabulary* v, const uint32_t* leaf_node, const uint32_t* level_node, int n, uint32_t* bow_ids,
                       double* bow_values, int* n_words, uint32_t* fv_nodes, uint32_t* fv_offsets, uint32_t* fv_features,
                       int* n_fv_nodes, uint32_t* word_of_feature);

/* int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  (ORBmatcher.cc:154-283;
 * strict_threshold = 0, valid2 = NULL) and int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2,
 * vector<MapPoint*>& vpMatches12)  (:517-650; strict_threshold = 1: `bestDist1 < TH_LOW`).
 * Side 1 is the keyframe whose MapPoints are searched for: valid1[i] != 0 iff keypoint i has a MapPoint that is
 * not bad; valid2 likewise for side 2 (NULL = every keypoint is a candidate, as for a Frame).  angle1 / angle2: the
 * keypoints' angles (needed when check_orientation).  fvX_*: the FeatureVectors in the layout
 * orbfe_bow_transform writes.  matches12[i1] = matched index on side 2 or -1 (capacity n1); for the Frame overload
 * vpMapPointMatches[matches12[i1]] = vpMapPointsKF[i1].  *nmatches = the function's return value.
 * desc1 / desc2 may point to host memory or to 16-byte aligned rows in the matcher's device memory (a resident frame's
 * descriptors, orbfe_frame_descriptors_device): such a side is not copied anywhere. */
int orbfe_search_by_bow(orbfe_matcher* m, const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                        const uint32_t* fv1_nodes, const uint32_t* fv1_offsets, const uint32_t* fv1_features, int n_fv1,
                        const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                        const uint32_t* fv2_nodes, const uint32_t* fv2_offsets, const uint32_t* fv2_features, int n_fv2,
                        float nnratio, int check_orientation, int strict_threshold, int32_t* matches12, int* nmatches);

/* The SearchByBoW loop of Tracking::Relocalization (src/Tracking.cc:1005-1030: one SearchByBoW(pKF, mCurrentFrame, ...) per
 * candidate keyframe) as one call: n_kf keyframes (side 1: arrays of per-keyframe pointers / sizes) against ONE frame (side 2),
 * one upload, one kernel launch over every (keyframe, common vocabulary node) pair, one download.  matches12[k] / nmatches[k]
 * are exactly what orbfe_search_by_bow returns for keyframe k. */
int orbfe_search_by_bow_batch(orbfe_matcher* m, int n_kf, const uint8_t* const* desc1, const float* const* angle1,
                              const uint8_t* const* valid1, const int* n1, const uint32_t* const* fv1_nodes,
                              const uint32_t* const* fv1_offsets, const uint32_t* const* fv1_features, const int* n_fv1,
                              const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const uint32_t* fv2_nodes,
                              const uint32_t* fv2_offsets, const uint32_t* fv2_features, int n_fv2, float nnratio,
                              int check_orientation, int strict_threshold, int32_t* const* matches12, int* nmatches);

/* int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, vector<pair<size_t,size_t>>&
 * vMatchedPairs)  (ORBmatcher.cc:652-804, with CheckDistEpipolarLine :135-152), from the epipole onwards: the caller
 * computes (ex, ey) (:657-666) and passes F12 row-major.  has_mpX[i] != 0: the keypoint already has a MapPoint and is
 * skipped (:708, :726).  kpsX_un = mvKeysUn; scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2
 * (nlevels2 <= 16).  pairs_out: (idx1, idx2) int32 pairs in ascending idx1, capacity n1 pairs; *nmatches = return
 * value = number of pairs.  (The reference never sets vbMatched2, so a keypoint of pKF2 may appear in several pairs.) */
int orbfe_search_for_triangulation(orbfe_matcher* m, const OrbfeKeyPoint* kps1_un, const uint8_t* desc1,
                                   const uint8_t* has_mp1, int n1, const uint32_t* fv1_nodes, const uint32_t* fv1_offsets,
                                   const uint32_t* fv1_features, int n_fv1, const OrbfeKeyPoint* kps2_un,
                                   const uint8_t* desc2, const uint8_t* has_mp2, int n2, const uint32_t* fv2_nodes,
                                   const uint32_t* fv2_offsets, const uint32_t* fv2_features, int n_fv2, const float F12[9],
                                   float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                   int nlevels2, int check_orientation, int32_t* pairs_out, int* nmatches);

/* void Frame::antidistorsionarProyeccionEquidistante(cv::Mat& puntos)  (src/Frame.cc:355-384): os1's
 * equidistant-fisheye keypoint undistortion (camera `modo: 1`), used by Frame::UndistortKeyPoints (:286-320) and
 * Frame::ComputeImageBounds (:322-353).  Host double-precision math on n (x, y) float pairs, in place; K is the
 * float intrinsic matrix {fx, fy, cx, cy}.  (Microseconds of scalar work per frame: it stays on the host.) */
int orbfe_undistort_equidistant(float* xy, int n, float fx, float fy, float cx, float cy);
/* The pinhole branch of the same two callers: cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK)
 * (src/Frame.cc:307, 339; camera `modo: 0` with Camera.k1 != 0).  dist = mDistCoef: k1 k2 p1 p2 [k3 [k4 k5 k6]]
 * (src/Tracking.cc:1221-1243), ndist 0..8.  Host double-precision math, in place. */
int orbfe_undistort_pinhole(float* xy, int n, float fx, float fy, float cx, float cy, const float* dist, int ndist);
/* void Frame::ComputeImageBounds(const cv::Mat& imLeft)  (src/Frame.cc:322-353): bounds = {mnMinX, mnMaxX, mnMinY,
 * mnMaxY}; camera_mode 0 = pinhole (undistorted corners when dist[0] != 0, else the image rectangle), 1 = os1's
 * equidistant fisheye. */
int orbfe_compute_image_bounds(int cols, int rows, int camera_mode, float fx, float fy, float cx, float cy, const float* dist,
                               int ndist, float bounds[4]);

/* Frame::GetFeaturesInArea (src/Frame.cc:209-262) evaluated by the GPU candidate kernel, for the
 * parity tests: indices in reference order.  out[cap]. */
int orbfe_debug_features_in_area(orbfe_matcher* m, const OrbfeKeyPoint* kps_un, int n, const float bounds[4],
                                 float x, float y, float r, int min_level, int max_level, int32_t* out, int cap,
                                 int* n_out);

/* ---------------------------------------------------------------------------------------------
 * ONE camera stream over several GPUs (the single-stream shape: "round-robin frames over GPUs with an in-order
 * completion queue" -- what the reference itself is: one Video thread, one Tracking thread, frames strictly in order,
 * src/main.cc:113-141, System.cc:115-152).  Batch k of the stream is extracted (and matched inside the batch) on device
 * device_ids[k mod n] by an ordinary single-device runner with `depth` batches in flight; orbfe_stream_multi_pop returns the batches
 * strictly in push order whatever order the devices finish in.  No collective, no device-to-device copy: the predecessor of a
 * batch's FIRST frame (the previous batch's last frame, extracted on another device) takes a host bounce and that one pair per batch
 * is matched by the host-array search on the batch's own device.  Results are identical to one orbfe_stream fed the same frames.
 * A device may appear several times in device_ids (more batches in flight on it).  With in_device_memory != 0 the frames of a
 * push must live on the device that push goes to: orbfe_stream_multi_device_of_next_push.
 * Everything else as orbfe_stream_*: pointers returned by _pop stay valid until the next _pop; _set_* only while nothing is in flight;
 * push from one thread, pop from one thread; never push more than `depth` + 2 batches PER DEVICE ahead of the pops (a device's runner
 * has depth + 4 result slots and finished batches wait in them for their turn).
 * ------------------------------------------------------------------------------------------- */
typedef struct orbfe_stream_multi orbfe_stream_multi;
int orbfe_stream_multi_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, const int* device_ids,
                              int n_devices, int batch, int depth, orbfe_stream_multi** out);
void orbfe_stream_multi_destroy(orbfe_stream_multi* s);
int orbfe_stream_multi_devices(const orbfe_stream_multi* s);
int orbfe_stream_multi_capacity(const orbfe_stream_multi* s);
int orbfe_stream_multi_device_of_next_push(const orbfe_stream_multi* s);
int orbfe_stream_multi_set_matching(orbfe_stream_multi* s, const float bounds[4], int window_size, float nnratio, int check_orientation);
int orbfe_stream_multi_set_blur_variant(orbfe_stream_multi* s, int variant);
int orbfe_stream_multi_push(orbfe_stream_multi* s, const uint8_t* const* gray, int in_device_memory, int rows, int cols, size_t stride_bytes);
int orbfe_stream_multi_pop(orbfe_stream_multi* s, const OrbfeKeyPoint** kps, const uint8_t** desc, const int** n_kps, const int32_t** matches12,
                           const int** nmatches);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H_ */
