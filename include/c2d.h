/*
 * c2d.h — C-ABI of the MI355X-native batched 2D SAT collision engine (libc2d.so).
 *
 * This is the drop-in boundary of the hot path.  The reference exposes no FFI
 * (SURVEY.md §8b): its boundary is the set of __device__/__global__ functions in
 * utils.cu plus the host loops in the three main()s.  Every entry point below
 * names the reference interface it replaces (file:line under /root/reference).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no C++/torch types;
 *   - return an int status (C2D_OK == 0, negative on error), never exit();
 *     the reference prints and exit()s (utils.cu:59-72);
 *   - buffers are caller-owned; pointers named d_* / "device" are HIP device
 *     pointers on the context's device, everything else is host memory;
 *   - work is enqueued on the caller's stream (a hipStream_t passed as void*,
 *     NULL = the default stream) and is asynchronous unless stated otherwise;
 *   - a c2d_ctx is bound to one device and owns a small device workspace (partial
 *     counts, adaptive-loop lists): use one ctx per device and per host thread.  Calls
 *     that use the workspace (c2d_mc_scenes; the SAT entry points when d_count != NULL)
 *     are ordered by their stream; issuing one on stream B while an earlier one on
 *     stream A has not finished returns C2D_ERR_UNSUPPORTED (use one ctx per stream).
 *     Streams stay the caller's: c2d keeps no stream handle beyond the call it was given to
 *     (the guard reads completion stamps that the kernels raise themselves, never the
 *     runtime's view of a remembered stream), so a stream may be destroyed by any means at
 *     any time after its calls were issued.  A NEW stream that the runtime creates at a
 *     destroyed stream's address is another stream to the guard, not the old one: it compares
 *     hipStreamGetId where the runtime has it (HIP >= 7.1).  On an older runtime — the HIP 7.0
 *     a PyTorch process holds — streams have no number: c2d_stream_destroy forgets the address
 *     of a stream it destroys with calls in flight, but a stream destroyed by OTHER means with
 *     c2d calls still in flight, whose address the runtime gives to a new stream that is used
 *     with the same ctx at once, passes for the old one there: drain such a stream first (or
 *     destroy it through c2d_stream_destroy).  While a stream is being captured into a
 *     graph the guard stands aside: the order of a graph's replays against other work on the
 *     same ctx is the caller's to arrange.  A launch that fails after it took its place in the
 *     guard's bookkeeping takes itself out again (the failing call drains its stream).
 *
 * Arithmetic contract (DESIGN.md §"Canonical arithmetic"): IEEE binary32,
 * round-to-nearest-even, no multiply-add contraction except where the spec
 * says fma; sin/cos/log are the c2d polynomial forms (bit-reproducible on any
 * IEEE machine), sqrt and divide are correctly rounded.  Booleans and hit
 * counts are therefore bit-exact against the CPU oracle in oracle/.
 *
 * Non-finite inputs.  The SAT entry points (c2d_sat_rect_pairs_*, c2d_sat_poly_pairs*,
 * c2d_rects_from_poses) are defined for EVERY bit pattern and follow the reference there
 * too: thrust::minmax_element (utils.cu:176-177) is comparison based, so on each axis a NaN
 * projection of a polygon's FIRST vertex stays that polygon's extreme — both comparisons
 * of utils.cu:178 are then false and the axis does not separate — while a NaN projection of
 * a later vertex is skipped; infinities are ordered like numbers.  Consequence: a pair with
 * a NaN in vertex 0 of either polygon reads "collide".  The Monte-Carlo entry points
 * (c2d_mc_pair, c2d_mc_scenes, c2d_sample_scenes) take table-driven scene parameters and
 * require them to be finite and below 1e15 in magnitude, and every length either zero or at
 * least 1e-15 (a product of two smaller lengths is denormal, and the shortcuts' margins are
 * relative rounding bounds); outside that domain a call still terminates and stays memory-safe,
 * and its hit counts follow the same rules (every certain-miss shortcut is switched off for
 * such a scene).
 */
#ifndef C2D_H_
#define C2D_H_

#include <stddef.h>
#include <stdint.h>

#include "utils.h"

#ifdef __cplusplus
extern "C" {
#endif

#define C2D_VERSION_MAJOR 0
#define C2D_VERSION_MINOR 6

/* ---- status codes ------------------------------------------------------ */
#define C2D_OK 0
#define C2D_ERR_INVALID_ARG (-1)  /* NULL pointer, bad size, bad vertex count ...        */
#define C2D_ERR_HIP (-2)          /* a HIP runtime call failed; see c2d_last_error()     */
#define C2D_ERR_NO_DEVICE (-3)    /* no usable gfx950 device / device index out of range */
#define C2D_ERR_NOMEM (-4)        /* device or host allocation failed                    */
#define C2D_ERR_UNSUPPORTED (-5)  /* argument combination outside the documented domain  */
#define C2D_ERR_DIST (-6)         /* RCCL / multi-GPU set-up or collective failed        */

typedef struct c2d_ctx c2d_ctx;
typedef void* c2d_stream; /* hipStream_t */

/* Fixed sampling schedule of the adaptive Monte-Carlo loop
 * (reference compute_collision_probability.cu:283-287, generate_dataset.cu:427-431):
 * batches of 1000 samples while n_samples < 20000, then batches of 100000. */
#define C2D_MC_SMALL_BATCH 1000
#define C2D_MC_LARGE_BATCH 100000
#define C2D_MC_SWITCH_AT 20000

typedef struct c2d_device_info {
    char name[128];
    char arch[64];
    int device;
    int compute_units;
    int wavefront_size;
    int lds_bytes_per_cu;
    size_t hbm_bytes;
    char pci_bus_id[32]; /* "0000:8b:00.0" (hipDeviceGetPCIBusId): which card of the node this ctx sits on */
} c2d_device_info;

/* ---- library / context --------------------------------------------------- */
int c2d_version(void); /* MAJOR*1000 + MINOR */
const char* c2d_status_string(int status);
/* Last error text of this ctx (HIP error string and call site); "" if none. */
const char* c2d_last_error(const c2d_ctx* ctx);
int c2d_device_count(int* count);
/* Replaces the implicit device-0 + default-stream set-up of the reference mains
 * (compute_collision_probability.cu:212-251). */
int c2d_ctx_create(int device, c2d_ctx** out);
int c2d_ctx_destroy(c2d_ctx* ctx);
/* c2d_device_info may grow at its end (0.5 added pci_bus_id), so it is filled through a call that is told how large the
 * CALLER's struct is: c2d_ctx_info_sized writes the first min(out_bytes, sizeof(c2d_device_info)) bytes of the current layout
 * and nothing beyond out_bytes.  Sources compiled against this header get it through the c2d_ctx_info macro below, with the
 * size of the struct they were compiled with.  The EXPORTED symbol c2d_ctx_info stays for binaries built against the 0.4
 * header, which declared the struct without pci_bus_id: it writes that layout only (C2D_DEVICE_INFO_BYTES_0_4 bytes: everything
 * up to and including hbm_bytes), never past the end of an old caller's struct. */
#define C2D_DEVICE_INFO_BYTES_0_4 (offsetof(c2d_device_info, hbm_bytes) + sizeof(size_t))
int c2d_ctx_info(const c2d_ctx* ctx, c2d_device_info* out);
int c2d_ctx_info_sized(const c2d_ctx* ctx, c2d_device_info* out, size_t out_bytes);
#define c2d_ctx_info(ctx, out) c2d_ctx_info_sized((ctx), (out), sizeof(c2d_device_info))
/* Argument errors that only the device can see (today: a polygon vertex count outside
 * 1..C2D_POLY_KMAX) are reported asynchronously: the kernel records them in a pinned word
 * of the ctx and the first c2d_stream_synchronize — or this call, for callers that
 * synchronise by other means — after the kernel finished returns C2D_ERR_INVALID_ARG once
 * (c2d_last_error() says what) and clears the record.  Returns C2D_OK if nothing is pending. */
int c2d_ctx_check_async(c2d_ctx* ctx);

/* ---- memory / stream plumbing ---------------------------------------------
 * Replace cudaMalloc / cudaMemcpy / cudaFree / cudaDeviceSynchronize in the
 * reference mains (compute_collision_probability.cu:212-248, :314-334, :367-377;
 * generate_dataset.cu:371-405, :461-482, :512-522).  Copies are asynchronous
 * on `stream`; call c2d_stream_synchronize before touching the host buffer. */
int c2d_malloc(c2d_ctx* ctx, void** d_ptr, size_t bytes);
int c2d_free(c2d_ctx* ctx, void* d_ptr);
/* Page-locked host memory: copies to / from it really are asynchronous (a copy that involves pageable
 * memory is staged by the runtime and may hold the calling thread until it is done), which is what lets a
 * driver overlap the host side of one batch with the GPU side of the next (csrc/host/driver_common.hpp). */
int c2d_malloc_host(c2d_ctx* ctx, void** h_ptr, size_t bytes);
int c2d_free_host(c2d_ctx* ctx, void* h_ptr);
int c2d_memset(c2d_ctx* ctx, void* d_ptr, int value, size_t bytes, c2d_stream stream);
int c2d_memcpy_h2d(c2d_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, c2d_stream stream);
int c2d_memcpy_d2h(c2d_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, c2d_stream stream);
int c2d_stream_create(c2d_ctx* ctx, c2d_stream* out);
int c2d_stream_destroy(c2d_ctx* ctx, c2d_stream stream);
int c2d_stream_synchronize(c2d_ctx* ctx, c2d_stream stream);

/* ---- geometry ---------------------------------------------------------------
 *
 * c2d_rects_from_poses: batched create_rect (utils.cu:119-130) followed by
 * rot_trans_rectangle (utils.cu:132-142): rectangle i has size w[i] x h[i],
 * is rotated by theta[i] about its centre and translated to (cx[i], cy[i]).
 * Inputs are 5 device planes f32[n]; output is 8 device planes f32[n] in the
 * reference's flat order x0,y0,x1,y1,x2,y2,x3,y3. */
int c2d_rects_from_poses(c2d_ctx* ctx, const float* d_cx, const float* d_cy, const float* d_w,
                         const float* d_h, const float* d_theta, size_t n,
                         float* const d_out_planes[8], c2d_stream stream);

/* c2d_sat_rect_pairs_verts: batched rectangle-rectangle SAT — the arithmetic
 * of convex_collide (utils.cu:159-184) applied to n independent pairs: 8 axes
 * (the edge vectors of both rectangles, utils.cu:170-171), all 4+4 vertices
 * projected with an unfused dot product (utils.cu:172-175), separated iff
 * max1 < min2 || max2 < min1 (strict, utils.cu:178), no early result change.
 * The reference only calls convex_collide from inside its MC kernel
 * (compute_collision_probability.cu:138); this entry point is that function
 * over SoA arrays.
 *   d_planes[0..7]  : rectangle 1, planes x0,y0,x1,y1,x2,y2,x3,y3, each f32[n]
 *   d_planes[8..15] : rectangle 2, same order
 *   d_out           : u8[n], 1 = collide, 0 = separated
 *   d_count         : optional (may be NULL) device uint64 that is *incremented*
 *                     by the number of colliding pairs (atomically; zero it first).
 * Any alignment is accepted; planes and d_out aligned to 16 B / 4 B take the
 * wide-load path. */
int c2d_sat_rect_pairs_verts(c2d_ctx* ctx, const float* const d_planes[16], size_t n,
                             uint8_t* d_out, unsigned long long* d_count, c2d_stream stream);

/* c2d_sat_rect_pairs_verts_mask: the same test with a bit mask as output, for callers that only
 * need collide / no-collide: bit (i & 63) of d_mask[i >> 6] is the result of pair i (1 = collide);
 * d_mask is a device u64[(n + 63) / 64], 8-byte aligned; the unused high bits of the last word are 0.
 * 64.125 instead of 65 bytes of traffic per pair. */
int c2d_sat_rect_pairs_verts_mask(c2d_ctx* ctx, const float* const d_planes[16], size_t n,
                                  unsigned long long* d_mask, unsigned long long* d_count,
                                  c2d_stream stream);

/* c2d_sat_rect_pairs_aos: the same test on the reference's own argument layout,
 * convex_collide(float* r1, float* r2) (utils.cu:159) with flat float[8]
 * rectangles, batched: d_r1 and d_r2 are f32[n][8] (16-byte aligned), pair i is
 * (d_r1 + 8*i, d_r2 + 8*i).  Same 65 B/pair as the plane format. */
int c2d_sat_rect_pairs_aos(c2d_ctx* ctx, const float* d_r1, const float* d_r2, size_t n,
                           uint8_t* d_out, unsigned long long* d_count, c2d_stream stream);

/* c2d_sat_rect_pairs_pose: the same test on pose-format input: for each pair,
 * the result is that of building both rectangles exactly as c2d_rects_from_poses would
 * (utils.cu:119-142) and testing them (utils.cu:159-184) — for every bit pattern.  The kernel
 * reaches it from the closed-form gap of the two rectangles wherever that gap exceeds a proven
 * rounding margin, and by that very vertex arithmetic elsewhere.
 *   d_pose_planes[0..4] : rectangle 1: cx, cy, w, h, theta   (f32[n] each)
 *   d_pose_planes[5..9] : rectangle 2: cx, cy, w, h, theta */
int c2d_sat_rect_pairs_pose(c2d_ctx* ctx, const float* const d_pose_planes[10], size_t n,
                            uint8_t* d_out, unsigned long long* d_count, c2d_stream stream);

/* c2d_sat_rect_pairs_verts_host / _pose_host: the same tests for batches that live in HOST memory — what the
 * reference does around its kernel with one blocking cudaMemcpy after the other
 * (compute_collision_probability.cu:270-274 up, :314-318 down) — as ONE synchronous call: whole planes go up,
 * the test runs, the booleans and the count come back, in chunks of 2^24 pairs so that a batch of any size
 * needs at most 1 GB of device memory (kept by the ctx from the first call on).  It runs at the rate of the
 * host-to-device link, which carries 64 and 40 bytes per pair and is 95 % of the call (pipelined forms were
 * measured and are slower: csrc/c2d_host.hip).  h_planes / h_pose_planes: host planes as for the device entry
 * points, pageable or page-locked; h_out: host u8[n]; h_count: optional host word that receives the number of
 * colliding pairs.  Uses the ctx's count workspace like the device entry points.  Returns when h_out is complete. */
int c2d_sat_rect_pairs_verts_host(c2d_ctx* ctx, const float* const h_planes[16], size_t n, uint8_t* h_out,
                                  unsigned long long* h_count);
int c2d_sat_rect_pairs_pose_host(c2d_ctx* ctx, const float* const h_pose_planes[10], size_t n, uint8_t* h_out,
                                 unsigned long long* h_count);

/* c2d_sat_poly_pairs: SAT for arbitrary convex polygons with up to
 * C2D_POLY_KMAX vertices.  Same projection / strict-< interval test as
 * utils.cu:172-180, but the axis of edge e is its true normal (-e.y, e.x):
 * the reference's edge-as-axis shortcut (utils.cu:170-171) is only valid for
 * rectangles.
 *   d_vx, d_vy : f32[2][C2D_POLY_KMAX][n]   (polygon, vertex, pair) — pair index fastest
 *   d_k        : u8[2][n]                    vertex counts, 1..C2D_POLY_KMAX
 *   d_out      : u8[n]
 * Padded vertex slots (index >= count) are never interpreted.  The call is asynchronous
 * and graph-capturable; vertex counts are checked on the device: a pair with a count
 * outside 1..C2D_POLY_KMAX gets result 0 and the error is reported by the next
 * c2d_stream_synchronize / c2d_ctx_check_async (see there). */
int c2d_sat_poly_pairs(c2d_ctx* ctx, const float* d_vx, const float* d_vy, const uint8_t* d_k,
                       size_t n, uint8_t* d_out, unsigned long long* d_count, c2d_stream stream);

/* c2d_sat_poly_pairs_rows: the same test on a layout with `rows` vertex rows per polygon instead of
 * C2D_POLY_KMAX — d_vx, d_vy : f32[2][rows][n], vertex counts 1..rows, 1 <= rows <= C2D_POLY_KMAX.
 * Batches of small polygons (triangles and quadrilaterals: rows = 4; up to octagons: rows = 8) take
 * a quarter or half of the memory and run a kernel instance sized for them (fewer registers, more
 * waves per SIMD, four or eight pairs per wave in the full evaluation); layouts of 9 .. 15 rows run as
 * one bin of the binned kernel below (its 12- and 16-row instances).  rows = C2D_POLY_KMAX is
 * c2d_sat_poly_pairs.  A batch whose pairs are ordered by vertex counts moves only the rows its waves
 * need (16-row instance: a wave skips every row above its largest count). */
int c2d_sat_poly_pairs_rows(c2d_ctx* ctx, const float* d_vx, const float* d_vy, const uint8_t* d_k,
                            size_t n, int rows, uint8_t* d_out, unsigned long long* d_count,
                            c2d_stream stream);

/* ---- binned polygon batches ---------------------------------------------------
 * The padded layout above moves 16 vertex rows per polygon whatever the polygons are: with
 * K ~ U{3..16} that is 259 bytes per pair for 155 bytes of real vertices, and no kernel can
 * avoid it — the pairs of a wave have unrelated counts, so every 64-byte segment of every row
 * holds a vertex somebody needs.  A caller that keeps its pairs in BINS — pairs grouped by the
 * size of their polygons, each bin a tight plane layout of its own — hands over the exact
 * bytes instead, and the bin's row counts are known before any count byte has been read:
 *
 *   bin:  rows_a, rows_b        vertex rows of polygon A / B in this bin, 1..C2D_POLY_KMAX
 *         n                     pairs in the bin (any number, < 2^32; every vertex plane — rows x stride
 *                               floats — must stay below 4 GiB: split larger bins)
 *         stride                elements between the vertex rows of a plane (0 = n; a multiple of 64
 *                               with 256-byte aligned planes keeps every row segment aligned)
 *         d_ax, d_ay            f32[rows_a][stride]  polygon A's vertices, pair index fastest
 *         d_bx, d_by            f32[rows_b][stride]  polygon B's
 *         d_ka, d_kb            u8[n] vertex counts 1..rows_a / 1..rows_b, or NULL = every
 *                               polygon of the bin has exactly rows_a / rows_b vertices
 *         d_out                 u8[n] results
 *
 * Same arithmetic and the same results as c2d_sat_poly_pairs for the same polygons (true normals,
 * unfused projections, strict <; SAT is symmetric in A and B, so a producer may swap the two
 * polygons of a pair to halve the number of bins).  Any number of bins (up to 65535), any
 * mix of row counts; ONE launch covers all of them.
 *
 *   c2d_poly_bins_create   validates the descriptors and uploads the launch table (synchronous;
 *                          the buffers stay the caller's and may be refilled between tests: the
 *                          table holds pointers and sizes, not data);
 *   c2d_sat_poly_pairs_binned   tests every pair of every bin: asynchronous on `stream`,
 *                          graph-capturable; vertex counts are checked on the device as for
 *                          c2d_sat_poly_pairs (pair reads 0, error at the next synchronise);
 *   c2d_poly_bins_from_padded   bins a padded batch (the layout of c2d_sat_poly_pairs_rows) on the
 *                          device: polygon sizes are rounded up to a multiple of `granularity`
 *                          rows (1 = one bin per (ka, kb), no padding at all; 4 = at most 16 bins),
 *                          the bins live in ONE device block owned by the handle.  This moves
 *                          every vertex once (about 1.3 ms per 1e7 pairs, what eight tests save against the padded layout;
 *                          profiles/notes_r04_bin_move.md): it pays when the
 *                          batch is tested more than once or as a converter for stored datasets;
 *                          a producer that can write bins directly should.  Synchronous; a vertex
 *                          count outside 1..rows is refused (C2D_ERR_INVALID_ARG, no handle).
 *   c2d_poly_bins_results  for a handle made by c2d_poly_bins_from_padded: the results in the
 *                          ORDER OF THE PADDED INPUT, u8[n] (asynchronous on `stream`);
 *   c2d_poly_bins_get      descriptor i of the handle (device pointers), for inspection: bin i of
 *                          c2d_poly_bins_create is the caller's bin i (a bin with n = 0 stays in the list and
 *                          takes no work). */
typedef struct c2d_poly_bin {
    uint32_t rows_a, rows_b;
    size_t n;
    size_t stride;   /* elements between consecutive vertex rows of a plane, >= n; 0 = n */
    const float* d_ax;
    const float* d_ay;
    const float* d_bx;
    const float* d_by;
    const uint8_t* d_ka;
    const uint8_t* d_kb;
    uint8_t* d_out;
} c2d_poly_bin;
typedef struct c2d_poly_bins c2d_poly_bins;
int c2d_poly_bins_create(c2d_ctx* ctx, const c2d_poly_bin* bins, size_t n_bins, c2d_poly_bins** out);
int c2d_poly_bins_from_padded(c2d_ctx* ctx, const float* d_vx, const float* d_vy, const uint8_t* d_k, size_t n,
                              int rows, int granularity, c2d_poly_bins** out, c2d_stream stream);
int c2d_poly_bins_destroy(c2d_ctx* ctx, c2d_poly_bins* bins);
size_t c2d_poly_bins_size(const c2d_poly_bins* bins);   /* number of bins  */
size_t c2d_poly_bins_pairs(const c2d_poly_bins* bins);  /* pairs in total  */
size_t c2d_poly_bins_bytes(const c2d_poly_bins* bins);  /* bytes one test moves: vertices, counts, results */
int c2d_poly_bins_get(const c2d_poly_bins* bins, size_t i, c2d_poly_bin* out);
int c2d_sat_poly_pairs_binned(c2d_ctx* ctx, const c2d_poly_bins* bins, unsigned long long* d_count,
                              c2d_stream stream);
int c2d_poly_bins_results(c2d_ctx* ctx, const c2d_poly_bins* bins, uint8_t* d_out, c2d_stream stream);

/* ---- random stream -----------------------------------------------------------
 * Replaces setup_kernel + curand_normal (utils.cu:111-117, :146-150).  The
 * generator is counter based: Philox4x32-10 with key = seed and subsequence =
 * scene_id — rocRAND's rocrand_state_philox4x32_10 engine.  No state array, no
 * set-up kernel, results independent of launch geometry and of how samples are
 * sharded over GPUs.
 *
 * Draw layout.  The samples of a stream are drawn in groups of four: sample s is
 * member j = s & 3 of group g = s >> 2, which owns blocks 8g .. 8g+5 of the
 * subsequence (rocRAND: rocrand_init(seed, scene_id, offset = 4 * (8g + b))):
 *   block 8g+0, word j                  radius word of the first Box-Muller pair (dx, dy)
 *   block 8g+1, word j                  angle word of that pair
 *   block 8g+2+(j>>1), words 2(j&1)..   radius, angle word of the second pair (dtheta, dw)
 *   block 8g+4+(j>>1), words 2(j&1)..   third pair (dh; its second normal is unused)
 * The draw ORDER per sample is the reference's (dx, dy, dtheta, dw, dh).  Grouping
 * the four radius words of four samples in one block is what lets the kernels
 * prove "certain miss" for most samples of a far scene at a quarter of a Philox
 * block per sample.
 *
 * c2d_philox_normals (parity/debug): for samples sample_begin .. +n writes the
 * five N(0,1) draws in the reference's order dx,dy,dtheta,dw,dh
 * (utils.cu:146-150) to d_normals[n][5] and, if d_raw != NULL, the six raw
 * 32-bit words in draw order to d_raw[n][6]. */
int c2d_philox_normals(c2d_ctx* ctx, uint64_t seed, uint64_t scene_id, uint64_t sample_begin,
                       size_t n, float* d_normals, uint32_t* d_raw, c2d_stream stream);

/* c2d_math_eval (parity/debug): evaluates one canonical math function of the
 * arithmetic contract on n inputs given as raw 32-bit patterns, so that tests
 * can compare the device implementation with the oracle bit for bit.
 *   C2D_MATH_LOG        out0 = log(x)                 x = float(bits) > 0, normal
 *   C2D_MATH_SINCOS     out0, out1 = sin(x), cos(x)   stands in for utils.cu:133-134
 *   C2D_MATH_SINCOS_U32 out0, out1 = sin, cos of 2*pi*bits/2^32
 *   C2D_MATH_SQRT       out0 = correctly rounded sqrt(x), x in {+-0} U [2^-96, 2^96]
 *   C2D_MATH_BOX_MULLER out0, out1 = the two normals of words (bits, ~bits * 2654435761) */
#define C2D_MATH_LOG 0
#define C2D_MATH_SINCOS 1
#define C2D_MATH_SINCOS_U32 2
#define C2D_MATH_SQRT 3
#define C2D_MATH_BOX_MULLER 4
int c2d_math_eval(c2d_ctx* ctx, int fn, const uint32_t* d_in_bits, size_t n, float* d_out0,
                  float* d_out1, c2d_stream stream);

/* ---- Monte-Carlo collision probability ---------------------------------------
 *
 * c2d_mc_pair: one scene, sample-parallel.  Replaces the body of
 * monte_carlo_sample_collision_dataset_uniform for a single data point
 * (compute_collision_probability.cu:119-139): robot = create_rect(robot_w,
 * robot_h) rotated by pose->theta and moved to pos (:132-133); obstacle =
 * create_rect(pose->width, pose->height) (:128); each sample perturbs the
 * obstacle with sample_rectangle (utils.cu:144-157) and tests it with
 * convex_collide (utils.cu:159-184).  Samples sample_begin .. sample_begin +
 * n_samples - 1 of stream (seed, scene_id) are evaluated; *d_hits (device
 * uint64) is incremented by the number of colliding samples.  Disjoint sample
 * ranges may run on different GPUs and be summed. */
int c2d_mc_pair(c2d_ctx* ctx, float robot_w, float robot_h, const Position* pos, const Pose* pose,
                const StdDev* std_dev, uint64_t seed, uint64_t scene_id, uint64_t sample_begin,
                uint64_t n_samples, unsigned long long* d_hits, c2d_stream stream);

/* c2d_mc_scenes: many scenes with the reference's adaptive stopping rule.
 * Replaces the host loop + kernel + thrust compaction of
 * compute_collision_probability.cu:276-332 (= generate_dataset.cu:420-479):
 * every scene is sampled in batches (C2D_MC_* schedule above); after each
 * batch the 95 % half-width calcSlack (utils.cu:186-196, with the int
 * overflow D1 fixed) is compared with bin_accuracy[getBin(p)]
 * (utils.cu:198-207, with the out-of-bounds read D2 fixed); a scene stops at
 * the first check that passes, or when n_samples >= max_samples.
 * Scene i uses random stream (seed, scene_id_base + i), so results do not
 * depend on batching, completion order or the number of GPUs.
 * The whole loop is enqueued on `stream` without any read-back (the schedule
 * state lives on the device); the call only synchronises when a host output
 * (total_samples, iterations) is requested.  Every step of the schedule — until
 * n_samples >= max_samples — is enqueued, 2 launches per step; steps after the last
 * scene finished retire at once (~5 us).  Schedules of more than 100 000 steps are
 * refused (C2D_ERR_INVALID_ARG): use larger batches.  With a host output requested the
 * call looks at the device state every 64 steps and stops enqueuing once no scene is left. */
typedef struct c2d_mc_scenes_args {
    const Pose* d_poses;          /* device Pose[num_poses]          (utils.cu:91-94)  */
    uint32_t num_poses;
    const StdDev* d_std_devs;     /* device StdDev[num_std_devs] — standard deviations,
                                     i.e. sqrt of variances.npy (ccp.cu:188-194)       */
    uint32_t num_std_devs;
    const PositionWithVarAndPoseIdx* d_scenes; /* device rows (x,y,var_idx,pose_idx)  */
    size_t n_scenes;
    float robot_w, robot_h;       /* ccp.cu:39-40 defaults 4.07 x 1.74                 */
    const float* accuracy_bins;   /* host f32[n_accuracy_bins], e.g. {0,.01,.1,1}      */
    const float* bin_accuracy;    /* host f32[n_accuracy_bins-1], e.g. {1e-4,1e-3,1e-2} */
    uint32_t n_accuracy_bins;     /* <= 16                                             */
    uint32_t max_samples;         /* ccp.cu:38 default 4000000                         */
    uint64_t seed;
    uint64_t scene_id_base;
    /* Sampling schedule; all three 0 = the reference default (C2D_MC_* above).  ztest.cu
     * uses a constant batch of 10000 (ztest.cu:332-339): small = large = 10000. */
    uint32_t schedule_small_batch;  /* batch size while n_samples < schedule_switch_at   */
    uint32_t schedule_large_batch;  /* batch size afterwards                             */
    uint32_t schedule_switch_at;
    uint32_t* d_hits;             /* device u32[n_scenes]  out: colliding samples      */
    uint32_t* d_n_used;           /* device u32[n_scenes]  out: samples drawn          */
    PoseCPVarAndPoseIdx* d_rows;  /* optional device rows (x,y,cp,var_idx,pose_idx) =
                                     one output .npy row each (ccp.cu:337-344), cp =
                                     hits / n_used (utils.cu:210-215)                  */
    uint64_t* total_samples;      /* optional host out: sum of n_used                  */
    uint32_t* iterations;         /* optional host out: schedule steps executed        */
} c2d_mc_scenes_args;

int c2d_mc_scenes(c2d_ctx* ctx, const c2d_mc_scenes_args* args, c2d_stream stream);

/* c2d_sample_scenes: draws the scenes themselves, replacing the iteration==0
 * branch of the generate_dataset kernel (generate_dataset.cu:207-219):
 * pose_idx and var_idx uniform over the tables, robot placed on a ring around
 * the obstacle (formula SURVEY.md §5.6).  Scene i uses stream
 * (seed, scene_id_base + i) in a key domain disjoint from the MC samples. */
int c2d_sample_scenes(c2d_ctx* ctx, const Pose* d_poses, uint32_t num_poses,
                      const StdDev* d_std_devs, uint32_t num_std_devs, float robot_w,
                      float robot_h, float spread, uint64_t seed, uint64_t scene_id_base,
                      size_t n_scenes, PositionWithVarAndPoseIdx* d_scenes, c2d_stream stream);

/* ---- Monte-Carlo collision probability for convex polygons -----------------------
 * The reference's README (README.md:3) says its code "can easily be extended to handle arbitrary
 * convex 2D shapes"; its own functions stop at rectangles (sample_rectangle utils.cu:144-157,
 * convex_collide utils.cu:159-184).  These two entry points are that extension of c2d_mc_pair /
 * c2d_mc_scenes, with the same random stream, draw order, sample sharding and stopping rule.
 * STATUS: c2d_mc_poly_* goes beyond the reference — the semantics below (notably "dw, dh scale the
 * obstacle frame") are this build's own generalisation, pinned only by this build's own oracle and by
 * the rectangle case; no reference code or fixture stands behind them.  They are an extension OUTSIDE
 * BASELINE.json's configs (none of the five names a polygon Monte-Carlo), frozen as of round 5: kept
 * working and tested, not developed further.
 *
 *   robot     a polygon in its own frame, rotated by theta and moved to pos with the arithmetic of
 *             rot_trans_rectangle (utils.cu:132-142; ccp.cu:132-133);
 *   obstacle  a polygon about the origin (ccp.cu:128).  A sample draws the five normals of
 *             utils.cu:146-150 in that order and applies them as sample_rectangle does: dw, dh change
 *             the SHAPE first — the obstacle frame's x / y coordinates are scaled by (1 + dw), (1 + dh),
 *             so StdDev.width / .height are RELATIVE standard deviations here (a w x h rectangle given
 *             as a 4-gon with sigma_w / w, sigma_h / h has the distribution of the reference's sample,
 *             utils.cu:152-155) — then the shape is rotated by dtheta about the origin and moved by
 *             (dx, dy) (utils.cu:156);
 *   test      the projection / strict-< interval test of utils.cu:172-180 on the true normals of all
 *             ka + kb edges, exactly as c2d_sat_poly_pairs.
 *
 * With sigma_w = sigma_h = 0 a rectangle given as a 4-gon gets, sample for sample, the very vertices
 * c2d_mc_pair gives it; the two tests then differ only in the scale of their axes (edge vector there,
 * normal here), which can move a boolean only for a sample within an ulp of touching.
 * Vertex order may be clockwise or counter-clockwise; 1 <= k <= C2D_POLY_KMAX (k = 1, 2: a point, a
 * segment).  The certain-miss shortcuts and the fast evaluation require finite parameters small enough
 * for no intermediate to overflow or to turn denormal: vertices, position, sigma_x, sigma_y zero or
 * between 1e-15 and 1e8 in magnitude, the
 * relative deviations sigma_w, sigma_h below 1e4 (a scale factor multiplies every coordinate), angles
 * below 1e15; outside that domain every sample is evaluated in full with the all-bit-patterns test and
 * the hit counts still equal the oracle's. */
typedef struct c2d_polygon {
    uint32_t k;                   /* vertices used */
    float x[C2D_POLY_KMAX];
    float y[C2D_POLY_KMAX];
} c2d_polygon;

/* c2d_mc_poly_pair: one polygon scene, sample-parallel (c2d_mc_pair's contract: samples sample_begin ..
 * sample_begin + n_samples - 1 of stream (seed, scene_id); *d_hits is incremented).  robot, obstacle,
 * pos and std_dev are host pointers read before the call returns. */
int c2d_mc_poly_pair(c2d_ctx* ctx, const c2d_polygon* robot, const Position* pos, float robot_theta,
                     const c2d_polygon* obstacle, const StdDev* std_dev, uint64_t seed, uint64_t scene_id,
                     uint64_t sample_begin, uint64_t n_samples, unsigned long long* d_hits, c2d_stream stream);

/* One entry of the polygon scene table: what Pose {width, height, theta} (utils.cu:91-94) is to the
 * rectangle dataset — the robot's rotation in the scene and the obstacle's shape. */
typedef struct c2d_poly_pose {
    float theta;
    c2d_polygon obstacle;
} c2d_poly_pose;

/* c2d_mc_poly_scenes: c2d_mc_scenes for polygon scenes.  `base` carries the scenes, tables of standard
 * deviations, schedule, stop rule, seeds and outputs exactly as for c2d_mc_scenes; its d_poses, num_poses,
 * robot_w and robot_h are ignored and replaced by d_poly_poses / num_poly_poses (device table indexed by
 * the rows' pose_idx) and the robot polygon (host pointer).  A vertex count outside 1..C2D_POLY_KMAX in the
 * device table is clamped and reported by the next c2d_stream_synchronize / c2d_ctx_check_async. */
typedef struct c2d_mc_poly_scenes_args {
    c2d_mc_scenes_args base;
    const c2d_polygon* robot;            /* host */
    const c2d_poly_pose* d_poly_poses;   /* device c2d_poly_pose[num_poly_poses] */
    uint32_t num_poly_poses;
} c2d_mc_poly_scenes_args;
int c2d_mc_poly_scenes(c2d_ctx* ctx, const c2d_mc_poly_scenes_args* args, c2d_stream stream);

/* ---- the dataset's tables -------------------------------------------------------
 * c2d_uniform_table_minstd: the table fill of generate_dataset.cu:279-332 on the device.  The reference
 * draws its variance and pose tables on the host from ONE std::default_random_engine (minstd_rand0,
 * default seed), row by row, dimension d uniform in [lo[d], hi[d]) through
 * std::uniform_real_distribution<float>, and uploads them; this writes d_out[rows][dims] with exactly those
 * floats (libstdc++'s: one engine call per float, value = float(x - 1) * 2^-31 * (hi - lo) + lo), engine
 * calls first_draw .. first_draw + rows * dims - 1 of that engine — the variances first (first_draw = 0),
 * then the poses (first_draw = 5 * num_variances), as the reference.  lo, hi: host float[dims], dims <= 8.
 * c2d_sqrt_f32: element-wise correctly rounded square root, d_out[i] = sqrt(d_in[i]) — the standard
 * deviations of the variance table (generate_dataset.cu:309-317, compute_collision_probability.cu:188-194). */
int c2d_uniform_table_minstd(c2d_ctx* ctx, float* d_out, size_t rows, int dims, const float* lo, const float* hi,
                             uint64_t first_draw, c2d_stream stream);
int c2d_sqrt_f32(c2d_ctx* ctx, const float* d_in, float* d_out, size_t n, c2d_stream stream);

/* ---- multi-GPU aggregation ----------------------------------------------------
 * New work (the reference is single-GPU, compute_collision_probability.cu:212-251): pairs,
 * scenes and Monte-Carlo sample ranges shard over the GPUs of a node with no exchange on
 * the data path (random streams are keyed by scene id and sample index, so the union of
 * the shards is bit-identical to a one-GPU run), and ONE sum-reduction of the 64-bit hit /
 * sample / histogram counters closes a run.  One process per GPU; the reduction is
 * ncclAllReduce(uint64, sum) of RCCL over xGMI, loaded (dlopen librccl.so.1) at the first
 * c2d_dist_* call only.
 *
 *   c2d_dist_unique_id : rank 0 creates the 128-byte communicator id (ncclGetUniqueId) and
 *                        hands it to the other ranks by any channel it has;
 *   c2d_dist_init      : collective over all ranks (ncclCommInitRank) on the ctx's device;
 *   c2d_dist_init_file : the same with the id exchanged through `path`: rank 0 writes the
 *                        file atomically, the others wait up to timeout_s seconds for it;
 *                        `path` must not exist beforehand (use a fresh name per run) and is
 *                        removed again once every rank has joined;
 *   c2d_dist_all_reduce_sum_u64 / c2d_dist_broadcast_u64 : in place on device words,
 *                        asynchronous on `stream` like every other entry point;
 *   c2d_dist_barrier   : a one-word all-reduce followed by a stream synchronise.
 *
 * c2d_dist_init, c2d_dist_barrier (hence c2d_dist_init_file) and c2d_dist_stream_synchronize run
 * under a watchdog: if the peers do not arrive within the time limit (timeout_s of
 * c2d_dist_init_file, 0 = the default; $C2D_DIST_TIMEOUT_S or 300 s otherwise) they return
 * C2D_ERR_DIST instead of blocking for ever.  A helper thread is then left inside RCCL / HIP:
 * c2d_dist_timed_out() says so, the communicator refuses further use, c2d_dist_destroy leaves it
 * alone, and the process should report the error and end with _exit() — the runtime's teardown in
 * a normal exit() would race that thread (the drivers do exactly this).
 * c2d_dist_all_reduce_sum_u64 / c2d_dist_broadcast_u64 only ENQUEUE; a caller that waits for them
 * with c2d_stream_synchronize blocks for ever if a peer died after the communicator was built —
 * c2d_dist_stream_synchronize is the same wait under the watchdog.
 *
 * c2d_dist_transport() names the transport: "rccl" — the only one this library contains.  A
 * separate test build (lib-rehearsal/libc2d.so, `make lib-rehearsal`) replaces it by a sum
 * through small files, which lets several ranks share one device (RCCL refuses that) so that
 * the N > 1 host logic can be rehearsed on a one-GPU box; it reports "file (rehearsal)". */
#define C2D_DIST_ID_BYTES 128
typedef struct c2d_dist c2d_dist;
int c2d_dist_unique_id(void* id_out /* [C2D_DIST_ID_BYTES] */);
int c2d_dist_init(c2d_ctx* ctx, int rank, int world_size, const void* id, c2d_dist** out);
int c2d_dist_init_file(c2d_ctx* ctx, int rank, int world_size, const char* path, double timeout_s,
                       c2d_dist** out);
int c2d_dist_rank(const c2d_dist* dist);
int c2d_dist_world_size(const c2d_dist* dist); /* as counted by RCCL (ncclCommCount) */
const char* c2d_dist_transport(const c2d_dist* dist);
/* Which RCCL sums the counters: ncclGetVersion's code (22203 = 2.22.3) and the file the library was loaded from (dladdr;
 * "" if unknown), so that a scaling record can be read without the logs.  Loads librccl if no c2d_dist_* call has yet.
 * Inside a process that already holds a librccl.so.1 — PyTorch ships its own — the loader hands back THAT one (same
 * soname), otherwise /opt/rocm's; both are plain RCCL and nothing else is ever used.  The rehearsal build reports
 * version 0 and "file (rehearsal)". */
int c2d_dist_rccl_version(int* version, char* path_out, size_t path_bytes);
int c2d_dist_all_reduce_sum_u64(c2d_dist* dist, unsigned long long* d_buf, size_t count,
                                c2d_stream stream);
int c2d_dist_broadcast_u64(c2d_dist* dist, unsigned long long* d_buf, size_t count, int root,
                           c2d_stream stream);
int c2d_dist_barrier(c2d_dist* dist, c2d_stream stream);
int c2d_dist_stream_synchronize(c2d_dist* dist, c2d_stream stream);
int c2d_dist_timed_out(const c2d_dist* dist); /* 1 after a watchdog time-out on this communicator */
int c2d_dist_destroy(c2d_dist* dist);

/* Host-side helpers with the reference's semantics, exported so that callers
 * and tests see exactly what the device evaluates (utils.cu:186-207). */
float c2d_calc_slack(uint32_t n_samples, uint32_t n_true);
int c2d_get_bin(float p, const float* accuracy_bins, uint32_t n_accuracy_bins);

#ifdef __cplusplus
}
#endif

#endif /* C2D_H_ */
