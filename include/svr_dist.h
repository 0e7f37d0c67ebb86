/* svr_dist.h — the sharded frame behind a C ABI: the multi-GPU form of VulkanEngine::draw() for a C++ host.
 *
 * The reference renders on one device (deviceIndex 0 / deviceMask 0, src/vk_engine.cpp:1295,1310) and
 * presents with vkQueuePresentKHR (:1332).  BASELINE.json's north_star shards the frame's tile rows over
 * the GPUs of a node and exchanges the finished rows over xGMI; this header is that form for a host that
 * stays C++: one process (or thread) per GPU, each with its own SvrContext (include/svr.h), scene
 * replicated, rank r rendering the rows [bounds[r], bounds[r+1]) under a scissor, the bands presented
 * (vkutil::copy_image, src/vk_engine.cpp:1276) and exchanged in place with RCCL — ncclAllGather for equal
 * bands, grouped ncclSend / ncclRecv for cost-balanced unequal ones — so that every rank ends up with the
 * whole presentable image.  Two frame slots (the reference keeps FRAME_OVERLAP = 3, src/vk_engine.h:77):
 * the exchange of frame i runs on its own stream while frame i + 1 renders.
 *
 * Call sequence per frame, at the site of draw_geometry (src/vk_engine.cpp:1265):
 *     svr_dist_begin_frame(d);                 // slot targets bound, scissor = this rank's band
 *     svr_clear_color / svr_draw_background;   // as before
 *     svr_draw_geometry(ctx, ...);             // as before
 *     svr_dist_end_frame(d);                   // present the band, start the exchange
 *     ... svr_dist_wait_frame(d, &image)       // where vkQueuePresentKHR was: the oldest frame in flight, whole
 * Built as simple-vk-renderer_amd/csrc/libsvr_dist.so (links libsvr_hip.so and librccl.so).
 * Python mirror with torch.distributed instead of raw RCCL: simple-vk-renderer_amd/dist.py.
 */
#ifndef SVR_DIST_H
#define SVR_DIST_H

#include "svr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SvrDist SvrDist;

#define SVR_DIST_ID_BYTES 128 /* sizeof(ncclUniqueId) */

enum SvrDistTransport {
  SVR_DIST_RCCL = 0, /* RCCL communicator over the node's GPUs (one rank per GPU) */
  SVR_DIST_SHM = 1   /* test transport: POSIX shared memory + host copies; any number of ranks on one GPU */
};

/* Rank 0 makes the id (RCCL: ncclGetUniqueId; SHM: the name of a shared-memory object) and hands it to the
 * other ranks by the application's own means (pipe, file, MPI, ...). */
int svr_dist_get_unique_id(int transport, uint8_t id[SVR_DIST_ID_BYTES]);

/* Collective: every rank calls it with the same id / world / extent.  ctx: the rank's context, created at the
 * full frame extent width x height (on the rank's GPU: SvrConfig.device).  swapchain_format: SVR_SWAPCHAIN_*.
 * Takes over the context's stream (svr_set_stream) and targets (svr_bind_targets) until svr_dist_destroy. */
int svr_dist_create(SvrContext* ctx, int transport, const uint8_t id[SVR_DIST_ID_BYTES], int rank, int world,
                    uint32_t width, uint32_t height, int swapchain_format, SvrDist** out);
void svr_dist_destroy(SvrDist* d);

/* The partition: world + 1 non-decreasing row numbers, bounds[0] = 0, bounds[world] = height; a band may be
 * empty.  Initially equal bands of ceil(height / world) rows.  set: every rank must set the same values
 * between the same two frames (it takes effect at the next svr_dist_begin_frame). */
int svr_dist_get_bounds(SvrDist* d, uint32_t* bounds, size_t capacity);
int svr_dist_set_bounds(SvrDist* d, const uint32_t* bounds, size_t count);

/* Collective (every rank, between the same two frames): re-cut the frame into bands of equal cost from the
 * ranks' tile-row costs (svr_get_row_costs), each band's costs scaled to measured_gpu_ms (this rank's GPU time
 * for its band — SvrStats.tile_ms rather than gpu_time_ms: geometry and binning are the same for every band, and
 * spread over a band's rows they make a short, dense band look dearer per row than it is; <= 0: the cost model alone).  One all-reduce of `height` 64-bit
 * sums; the cut itself (bottleneck-optimal, integer arithmetic) is computed identically on every rank.
 * *changed (may be NULL) = 1 when the boundaries moved. */
int svr_dist_rebalance(SvrDist* d, float measured_gpu_ms, int* changed);

/* this rank's band under the current partition (what the next svr_dist_begin_frame will use): a rank whose band is
 * empty (*rows = 0) skips its draw calls for the frame, but still begins and ends it */
int svr_dist_band(SvrDist* d, uint32_t* first_row, uint32_t* rows);

/* How the frame's rows are dealt out (SURVEY.md section 8e names both and says: measure).
 *   SVR_DIST_BANDS        rank r renders the contiguous rows [bounds[r], bounds[r+1]) (above): every rank repeats only the
 *                         geometry that can reach its band, but a band of few tile rows lasts as long as its deepest tile.
 *   SVR_DIST_INTERLEAVED  rank r renders the 32-row tile rows t with t % world == r (svr_set_row_interleave): deep and
 *                         shallow regions are dealt out evenly, every rank runs the whole scene's vertex stage.  The rows
 *                         travel as one RCCL group of in-place all-gathers, one per `world` consecutive tile rows.
 * set: every rank, between the same two frames; takes effect at the next svr_dist_begin_frame.
 * pick: collective; every rank hands in what a frame cost it under either partition (GPU ms, measured over a few
 * frames each); the partition whose SLOWEST rank is faster is chosen on every rank alike. */
enum SvrDistPartition { SVR_DIST_BANDS = 0, SVR_DIST_INTERLEAVED = 1 };
int svr_dist_set_partition(SvrDist* d, int partition);
int svr_dist_get_partition(SvrDist* d, int* partition);
int svr_dist_pick_partition(SvrDist* d, float bands_ms, float interleaved_ms, int* picked);

int svr_dist_begin_frame(SvrDist* d);
int svr_dist_end_frame(SvrDist* d);
/* The oldest frame whose exchange has been started and not yet waited for: blocks until every band has
 * arrived; *image_dev = device pointer of the whole width x height x 4-byte image (valid until the slot's next
 * svr_dist_begin_frame).  SVR_ERR_INVALID_ARGUMENT when no frame is in flight. */
int svr_dist_wait_frame(SvrDist* d, const void** image_dev);
/* A pass that overflows the renderer's internal queues is void and replayed later (include/svr.h, SVR_OPT_QUEUE_CAPS),
 * the present behind it too — but the exchange is this library's own stream work, which the replay knows nothing of: the
 * rows that travelled were stale.  Every present therefore reports into a status word (svr_set_present_status) that
 * travels with the rows; svr_dist_wait_frame finds a rank's word raised — on every rank alike, no extra message —
 * fences the renderer (the replay runs there) and exchanges the slot again before it hands the image out: a frame it
 * returns is a finished one (src/vk_engine.cpp:1226, 1332).  *n = frames that were exchanged a second time. */
int svr_dist_replays(SvrDist* d, uint32_t* n);
/* svr_dist_wait_frame + copy to host memory */
int svr_dist_read_frame(SvrDist* d, void* dst_host, size_t bytes);

const char* svr_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SVR_DIST_H */
