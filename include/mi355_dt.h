/*
 * mi355_dt.h -- C ABI of libmi355_dt.so: the MI355X (gfx950) detect-and-track
 * hot path of ktzsh/object-tracking.
 *
 * The reference has no FFI on this path: its hot path is Keras calls made from
 * two Python classes.  The in-tree precedent for
 * "Python host <-> C-ABI .so" is models_detection/YOLO.py:6-37,58-119 (ctypes
 * over libdarknet.so).  Each entry point below names the reference call it
 * replaces.  Conventions:
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from dt_last_error(ctx);
 *   - pointers named d_* are DEVICE pointers (e.g. torch tensor.data_ptr());
 *     pointers named h_* are HOST pointers; the caller owns all of them;
 *   - all tensors are dense float32 NHWC unless stated; frames may be uint8;
 *   - every launch goes to the stream set with dt_set_stream (default: the
 *     null stream); calls are asynchronous with respect to the host unless
 *     stated; one dt_ctx per GPU per process, not re-entrant across threads;
 *   - there is NO CPU fallback: without a gfx950 device every compute entry
 *     point fails with DT_ERR_DEVICE.
 */
#ifndef MI355_DT_H
#define MI355_DT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported from a library built with -fvisibility=hidden */
#define DT_API __attribute__((visibility("default")))

#define DT_OK 0
#define DT_ERR_ARG 1      /* bad argument / shape */
#define DT_ERR_DEVICE 2   /* HIP error (no device, launch failure, OOM) */
#define DT_ERR_STATE 3    /* weights not loaded / wrong call order */

#define DT_FRAMES_U8 0    /* uint8 HWC frames; x/255. fused into conv_1 (utils.py:150-153) */
#define DT_FRAMES_F32 1   /* float32 frames, already normalised */

#define DT_BOX_FLOATS 8   /* x, y, w, h, conf, label, score, cell */

typedef struct dt_ctx dt_ctx;

/* ---- context ---------------------------------------------------------- */
DT_API int dt_create(dt_ctx **out);
DT_API void dt_destroy(dt_ctx *ctx);
DT_API const char *dt_last_error(dt_ctx *ctx);
DT_API int dt_set_stream(dt_ctx *ctx, void *hip_stream);
/* ABI version of this header: major*100+minor (1.07: 1.06 + dt_gemm_split, dt_policy_set) */
DT_API int dt_abi_version(void);

/* ---- detector: KerasYOLO ---------------------------------------------- */
/* Replaces KerasYOLO.load_model graph construction (KerasYOLO.py:277-405) for
 * image_h x image_w inputs (multiples of 32), nb_box anchors per cell and
 * nb_class classes; anchors[2*nb_box] as KerasYOLO.ANCHORS (KerasYOLO.py:45). */
DT_API int dt_detector_config(dt_ctx *ctx, int image_h, int image_w, int nb_box,
                       int nb_class, const float *h_anchors);

/* Replaces init_weights + WeightReader (KerasYOLO.py:244-274,
 * utility/utils.py:138-148).  h_blob = float32 contents of a darknet .weights
 * file INCLUDING its 4-float header (the reader's offset starts at 4).  Folds
 * inference BatchNorm (eps 1e-3) into each kernel and uploads.  *consumed
 * receives the reader offset after conv_23 (may be NULL). */
DT_API int dt_load_darknet_weights(dt_ctx *ctx, const float *h_blob, size_t n_floats,
                            size_t *consumed);

/* Replaces model.predict([input_image, dummy])  (KerasYOLO.py:531) and the
 * two-output detector Model of MultiObjDetTracker.py:162-164.
 *   d_frames  [batch, H, W, 3] uint8 or float32 (frames_dtype)
 *   d_netout  [batch, G, G, nb_box*(5+C)]  raw conv_23 output     (may be NULL)
 *   d_feat    [batch, G, G, 1024]          'conv_feat' activation (may be NULL) */
DT_API int dt_detect_forward(dt_ctx *ctx, const void *d_frames, int frames_dtype,
                      int batch, float *d_netout, float *d_feat);

/* Replaces KerasYOLO.extract's intermediate_layer_model for the named taps the
 * reference reaches for: "conv_23", "conv_feat", "act_13" (26x26x512 skip
 * tensor, the TinyTracker feature layer, config.json:9).  Valid after a
 * dt_detect_forward of the same batch; copies into d_out. */
DT_API int dt_detector_tap(dt_ctx *ctx, const char *name, int batch, float *d_out);

/* Replaces KerasYOLO.extract's intermediate_layer_model for ANY layer of the detector graph
 * (KerasYOLO.py:509-520: Model(inputs, self.model.get_layer(layer).output).predict(...)).  Layer names as the
 * reference / Keras give them: conv_1..conv_23 (Conv2D output; conv_23 incl. its bias), norm_1..norm_22
 * (BatchNormalization output), leaky_re_lu_1..21 (alias act_N) and conv_feat (after LeakyReLU),
 * max_pooling2d_1..5, lambda_1 (space_to_depth), concatenate_1, reshape_1 / lambda_2 (= conv_23 values).
 * The fused production path never materialises most of these: the call re-runs the graph up to the layer and
 * executes that layer un-fused (a one-image debugging call, not a hot path).
 *   shape4 [4] receives (batch, h, w, channels) of the layer (may be NULL); with d_out == NULL only the shape
 *   is returned; d_out holds out_floats floats and receives the dense NHWC tensor. */
DT_API int dt_detector_extract(dt_ctx *ctx, const void *d_frames, int frames_dtype, int batch,
                               const char *layer, float *d_out, size_t out_floats, int *shape4);

/* ---- frame ingest ----------------------------------------------------- */
/* Replaces cv2.resize(image, (IMAGE_H, IMAGE_W)) on decoded uint8 frames
 * (KerasYOLO.py:526, MultiObjDetTracker.py:302); the /255. of normalize() is fused into
 * conv_1.  OpenCV's 8-bit INTER_LINEAR scheme (half-pixel centres, 11-bit coefficients).
 *   d_src [n, src_h, src_w, 3] uint8  ->  d_dst [n, dst_h, dst_w, 3] uint8 */
DT_API int dt_ingest_resize(dt_ctx *ctx, const uint8_t *d_src, int n, int src_h, int src_w,
                     uint8_t *d_dst, int dst_h, int dst_w);

/* ---- decode_netout + NMS (utility/utils.py:208-257) -------------------- */
/*   d_netout  [batch, GH, GW, NB, 5+NC]  raw logits (NOT modified)
 *   d_boxes   [batch, cap, DT_BOX_FLOATS] surviving boxes in (row,col,b) order; rows from
 *             min(count, cap) on are zero-filled (the buffer may be uninitialised)
 *   d_counts  [batch] int32: number of survivors (may exceed cap; extra dropped)
 *   d_classes [batch, cap, NC] post-NMS class scores per box (may be NULL)
 *   d_post    [batch, GH, GW, NB, 5+NC] the reference's in-place-mutated netout
 *             (conf / thresholded class scores after NMS) (may be NULL)        */
DT_API int dt_decode(dt_ctx *ctx, const float *d_netout, int batch, int GH, int GW,
              int NB, int NC, float obj_threshold, float nms_threshold,
              const float *h_anchors, int cap, float *d_boxes, int *d_counts,
              float *d_classes, float *d_post);

/* The same with one (obj_threshold, nms_threshold) pair PER FRAME -- d_thresholds [batch, 2] float32 on the
 * device: streams of a batch may run different operating points (addition; the reference decodes one frame per
 * call with one pair, utils.py:208). */
DT_API int dt_decode_per_frame(dt_ctx *ctx, const float *d_netout, int batch, int GH, int GW,
                        int NB, int NC, const float *d_thresholds, const float *h_anchors, int cap,
                        float *d_boxes, int *d_counts, float *d_classes, float *d_post);

/* bbox_iou (utility/utils.py:155-173) on n pairs: d_pairs [n,8] -> d_iou [n] */
DT_API int dt_bbox_iou(dt_ctx *ctx, const float *d_pairs, int n, float *d_iou);

/* ---- tracker: MultiObjDetTracker --------------------------------------- */
/* Replaces MultiObjDetTracker.load_model's recurrent head
 * (MultiObjDetTracker.py:175-183) + load_weights (:291-293).  Keras layouts:
 *   h_kernel     [3,3,Cb+1024,4U]  ConvLSTM2D input kernel, x_bbox channels first
 *   h_recurrent  [3,3,U,4U]        recurrent kernel
 *   h_bias       [4U]              gate order i,f,c,o
 *   h_out_kernel [1,1,U,Cb]        'tconv_2' 1x1 conv,  h_out_bias [Cb]
 * with Cb = nb_box*(5+C), U = units (512). */
DT_API int dt_tracker_load(dt_ctx *ctx, int units, const float *h_kernel,
                    const float *h_recurrent, const float *h_bias,
                    const float *h_out_kernel, const float *h_out_bias);

/* Replaces model.predict([x, b]) of the tracker model (MultiObjDetTracker.py:307).
 *   d_frames [n_clips, T, H, W, 3]
 *   d_trk    [n_clips, T, G, G, Cb]  'tracking' output grid   (may be NULL)
 *   d_det    [n_clips, T, G, G, Cb]  'detection' output grid  (may be NULL)
 * ConvLSTM state is zero at t=0 of every call (stateless Keras RNN). */
DT_API int dt_track_forward(dt_ctx *ctx, const void *d_frames, int frames_dtype,
                     int n_clips, int T, float *d_trk, float *d_det);

/* The two halves of dt_track_forward, exposed so that ONE stream can use several GPUs (frame-shard, SURVEY.md 8e row 3,
 * BASELINE.json configs[4]): each rank runs the TimeDistributed detector (MultiObjDetTracker.py:166-171) on its share
 * of the frames, the per-frame rows z = [conv_feat 1024 | x_bbox Cb | zero pad] are exchanged (RCCL all-gather), and
 * the owner of a clip runs ConvLSTM2D + tconv_2 (:175-183) on the stitched rows.  dt_track_row_width = floats per grid
 * cell of a row (1024 + Cb rounded up to 32).
 *   dt_track_detect:    d_frames [n_frames,H,W,3]      -> d_z [n_frames, G, G, row_width]
 *   dt_track_recurrent: d_z [n_clips, T, G, G, row_width] -> d_trk [n_clips,T,G,G,Cb] (+ d_det, may be NULL)
 * dt_track_forward == dt_track_detect on all frames followed by dt_track_recurrent. */
DT_API int dt_track_row_width(dt_ctx *ctx);
DT_API int dt_track_detect(dt_ctx *ctx, const void *d_frames, int frames_dtype, int n_frames, float *d_z);
DT_API int dt_track_recurrent(dt_ctx *ctx, const float *d_z, int n_clips, int T, float *d_trk, float *d_det);

/* The same split ONE STEP LATER in the graph (the default of parallel.track_clips_frame_sharded): ConvLSTM2D's input
 * projection W * x_t + b (MultiObjDetTracker.py:176) does not depend on the recurrence, so the rank that ran the detector on a
 * frame runs it as well -- 55 % of the recurrent head's FLOPs move from the clip's owner (sequential in T) to the part that is
 * spread over all GPUs -- and the rows that travel are the projection rows, dt_track_xproj_width = 4 * units floats per grid cell.
 *   dt_track_detect_xproj:    d_frames [n_frames,H,W,3] -> d_xp [n_frames, G, G, 4U]  (+ d_det [n_frames,G,G,Cb], may be NULL)
 *   dt_track_recurrent_xproj: d_xp [n_clips, T, G, G, 4U] -> d_trk [n_clips,T,G,G,Cb]
 * dt_track_forward == dt_track_detect_xproj on all frames followed by dt_track_recurrent_xproj. */
DT_API int dt_track_xproj_width(dt_ctx *ctx);
DT_API int dt_track_detect_xproj(dt_ctx *ctx, const void *d_frames, int frames_dtype, int n_frames, float *d_xp, float *d_det);
DT_API int dt_track_recurrent_xproj(dt_ctx *ctx, const float *d_xp, int n_clips, int T, float *d_trk);

/* Track identity (BUILD-DEFINED, DESIGN.md "Track identity"; the reference has
 * none).  d_boxes [n_clips,T,cap,8], d_counts [n_clips,T]
 * -> d_ids [n_clips,T,cap] int32 (-1 unused), d_nids [n_clips] ids opened. */
DT_API int dt_associate(dt_ctx *ctx, const float *d_boxes, const int *d_counts,
                 int n_clips, int T, int cap, float assoc_threshold,
                 int *d_ids, int *d_nids);

/* ---- cross-stream detection exchange (multi-GPU; the reference has no counterpart) ---- *
 * north_star: "RCCL all-gather of detections over xGMI only for cross-stream association".  Each rank packs its
 * detection table (the outputs of dt_decode + dt_associate for its clips) into ONE int32 row per clip,
 *     row = [ boxes T*cap*8 (float bits) | ids T*cap | counts T | nids | valid ],   dt_packed_row_ints(T, cap) ints,
 * padded with empty rows (valid = 0) to n_rows = the largest clip count of any rank, so that the exchange is ONE
 * fixed-size all-gather per step (the caller's collective: ncclAllGather / torch.distributed on d_rows, n_rows *
 * row ints per rank).  dt_unpack_detections compacts the gathered rows of all ranks (rank-major = global clip order)
 * back into tables and makes the per-clip track ids globally unique: gid = id + sum of nids of all earlier clips.
 *   dt_pack_detections:   d_boxes [n_clips,T,cap,8], d_counts [n_clips,T], d_ids [n_clips,T,cap], d_nids [n_clips]
 *                         -> d_rows [n_rows, row] int32          (n_rows >= n_clips; n_clips may be 0)
 *   dt_unpack_detections: d_rows [n_rows, row] -> d_boxes/d_counts/d_ids/d_nids sized for n_rows clips (the first
 *                         *d_n_valid are filled), d_gids [n_rows,T,cap] int64 (may be NULL), d_n_valid [1] int32 */
DT_API size_t dt_packed_row_ints(int T, int cap);
DT_API int dt_pack_detections(dt_ctx *ctx, const float *d_boxes, const int *d_counts, const int *d_ids,
                       const int *d_nids, int n_clips, int T, int cap, int n_rows, int32_t *d_rows);
DT_API int dt_unpack_detections(dt_ctx *ctx, const int32_t *d_rows, int n_rows, int T, int cap, float *d_boxes,
                         int *d_counts, int *d_ids, int *d_nids, int64_t *d_gids, int *d_n_valid);

/* ---- TinyTracker (models_tracking/TinyTracker.py:25-41) ---------------- */
/* Also serves TinyHeatmapTracker (models_tracking/TinyHeatmapTracker.py:26-48): same
 * graph with a heatmap_size^2-wide detection input and Dense(heatmap_size^2, sigmoid).
 *   h_kernel [D,4U], h_recurrent [U,4U], h_bias [4U], h_dense_kernel [U,out_dim],
 *   h_dense_bias [out_dim]; D = pooled feature width + detection input width
 *   (4 or heatmap_size^2); out_dim = 4 or heatmap_size^2.  pool: 0 = 'Global', 1 = 'Max'. */
DT_API int dt_tiny_load(dt_ctx *ctx, int D, int units, int out_dim, const float *h_kernel,
                 const float *h_recurrent, const float *h_bias,
                 const float *h_dense_kernel, const float *h_dense_bias);

/* Replaces model_tracker.predict([img_fv, det]).
 *   d_feat [n_seq, T, fh, fw, fc], d_det [n_seq, T, D - feature width] -> d_out [n_seq, T, out_dim] */
DT_API int dt_tiny_forward(dt_ctx *ctx, const float *d_feat, const float *d_det,
                    int n_seq, int T, int fh, int fw, int fc, int pool,
                    float *d_out);

/* The two halves of dt_tiny_forward, exposed so that a frame-sharded deployment
 * (BASELINE.json configs[3]) can all-gather the small per-frame rows between them:
 *   dt_tiny_features: pool + concatenate([feat, det]) -> d_x [n_rows, D]   (TinyTracker.py:29-34)
 *   dt_tiny_sequence: d_x [n_seq, T, D] -> LSTM over T -> Dense -> d_out [n_seq, T, 4]  (:36-37) */
DT_API int dt_tiny_features(dt_ctx *ctx, const float *d_feat, const float *d_det, int n_rows,
                     int fh, int fw, int fc, int pool, float *d_x);
DT_API int dt_tiny_sequence(dt_ctx *ctx, const float *d_x, int n_seq, int T, float *d_out);

/* generate_heatmap_feat (utility/utils.py:53-58) applied as the data generator does
 * (preprocessing.py:455): d_box4 [n,4] centre-format (cx,cy,w,h) -> d_heat [n, hs*hs] of 0/1.
 * generate_rectangle_from_heatmap (utility/utils.py:61-79): d_heat [n, hs*hs] ->
 * d_rect [n,4] int32 (x1,y1,x2,y2), (hs,hs,-1,-1) when no cell reaches thresh. */
DT_API int dt_heatmap_from_boxes(dt_ctx *ctx, const float *d_box4, int n, int hmap_size, float *d_heat);
/* same with the function's own argument list: d_xywh [n,4] float64 (det_x, det_y, det_w, det_h) */
DT_API int dt_heatmap_from_xywh64(dt_ctx *ctx, const double *d_xywh, int n, int hmap_size, float *d_heat);
DT_API int dt_rect_from_heatmap(dt_ctx *ctx, const float *d_heat, int n, int hmap_size, float thresh,
                         int *d_rect);

/* Detection box handed to the single-object tracker at inference (build-defined:
 * the reference only has the training-time choice, preprocessing.py:421-456):
 * highest-score survivor per frame as (cx,cy,w,h), zeros when a frame has none.
 *   d_boxes [n_frames, cap, 8], d_counts [n_frames] -> d_out4 [n_frames, 4] */
DT_API int dt_top_box(dt_ctx *ctx, const float *d_boxes, const int *d_counts, int n_frames,
               int cap, float *d_out4);

/* ---- training-target encoding (the step before the path when fine-tuning) ----
 * Replaces the object-coordinate fix at the end of BaseBatchGenerator.aug_image
 * (utility/preprocessing.py:171-188) + BatchGenerator.output_from_instance's y / b construction
 * (:214-293).  float64 like the reference's Python floats; bit-exact.
 *   d_objs   [n_frames, cap, 5] int32   xmin, ymin, xmax, ymax, LABELS index (-1: name not in LABELS)
 *   d_counts [n_frames] int32           objects per frame
 *   d_dims   [n_frames, 2] int32        original image (width, height)
 *   d_aug    [n_frames, 4] float64      the augmentation draw (scale, offx, offy, flip) of
 *                                       aug_image :150-166, or NULL for augment=False
 *   h_anchors [2*nb_box] float64 (host) ANCHORS
 *   d_y [n_frames, grid_h, grid_w, nb_box, 5+nb_class] float64, d_b [n_frames, true_box_buffer, 4] float64
 *   (b is the reference's (1,1,1,TRUE_BOX_BUFFER,4) block without the unit axes) */
DT_API int dt_encode_targets(dt_ctx *ctx, const int *d_objs, const int *d_counts, const int *d_dims,
                      const double *d_aug, int n_frames, int cap, int grid_h, int grid_w, int nb_box,
                      int nb_class, int image_h, int image_w, int true_box_buffer,
                      const double *h_anchors, double *d_y, double *d_b);

/* ---- hipGraph replay (low-latency serving; default off) ---------------- *
 * With graphs on, the launch-bound inner sequences that only touch library-owned buffers -- the detector
 * trunk conv_2..conv_21 and the ConvLSTM recurrence -- are captured once per shape (on the second call
 * with that shape) and replayed with hipGraphLaunch on an internal stream that is ordered after and
 * before the caller's stream with events.  Results are identical.  Graphs are dropped when weights are
 * reloaded or a workspace grows; profiling (dt_profile_enable) bypasses them. */
DT_API int dt_graph_enable(dt_ctx *ctx, int on);

/* ---- layer-level entry points (used by the parity tests) --------------- */
/* Conv2D 'same' stride 1 (+ optional folded bias, LeakyReLU slope, fused 2x2
 * maxpool) through the MFMA implicit-GEMM kernel.  h_kernel is Keras HWIO
 * [k,k,Cin,Cout]; Cin must be a multiple of 32 (use dt_conv1 for RGB input).
 *   pool: 0 none, 1 pooled output only, 2 both (d_out unpooled, d_out2 pooled) */
DT_API int dt_conv2d(dt_ctx *ctx, const float *d_in, int B, int H, int W, int Cin,
              const float *h_kernel, int k, int Cout, const float *h_bias,
              float leaky_slope, int pool, float *d_out, float *d_out2);

/* One ConvLSTM2D step (MultiObjDetTracker.py:176): d_x [B,H,W,Cx] with Cx a
 * multiple of 32, d_h/d_c [B,H,W,U] -> d_h_out/d_c_out. */
DT_API int dt_convlstm_step(dt_ctx *ctx, const float *d_x, int B, int H, int W, int Cx,
                     const float *d_h, const float *d_c, int U,
                     const float *h_kernel, const float *h_recurrent,
                     const float *h_bias, float *d_h_out, float *d_c_out);

/* The split-bf16 batched GEMM of the F(6x6,3x3) / F(4x4,3x3) layers (wino_gemm_s3.hip) on the caller's fp32 operands --
 * the contraction over input channels that models_detection/KerasYOLO.py:351-396 (conv_14 .. conv_22) and
 * models_tracking/MultiObjDetTracker.py:176 (both ConvLSTM2D convolutions) become in Winograd form -- so that the parity
 * tests can check the kernel against float64 AT THE SHAPES bench.py runs (P = 64, Mt = 7840, K = 1024 / 1280, ...):
 *   d_v [P][Mt][K], d_u [P][N][K] float32  ->  d_m [P][Mt][N] = sum_k v * u     (K % 32 == 0, N % 128 == 0)
 * Both operands are split into three bf16 terms on the device by the production pack kernel, then the production
 * launcher runs.  half: 0 = the launcher's own choice of row tile, 1 = 128-row tiles (two workgroups per CU), -1 = 256;
 * 2 = the 1x1 layers' form (P = 1): d_v is read as fp32 rows by the kernel itself, which splits its A fragments in registers. */
DT_API int dt_gemm_split_bf16(dt_ctx *ctx, const float *d_v, const float *d_u, int P, int Mt, int K, int N, int half,
                              float *d_m);
/* The same with the operand form chosen: nt = 3 -- three bf16 terms, six partial products per multiply (dt_gemm_split_bf16);
 * nt = 2 -- the fp16 form the library runs by default since round 6: each operand scaled by a power of two from its measured
 * max |x| (d_v as ONE tensor, like an activation; d_u per plane p, like the Winograd-domain weights) and carried as two fp16
 * terms hi + lo, three partial products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16, the fp32 accumulator scaled back
 * in the epilogue (P <= 64). */
DT_API int dt_gemm_split(dt_ctx *ctx, const float *d_v, const float *d_u, int P, int Mt, int K, int N, int half, int nt,
                         float *d_m);

/* ---- tuning / test knobs ------------------------------------------------- *
 * The DT_* environment variables of DESIGN.md's appendix are read ONCE, in dt_create (no launch path calls
 * getenv); this re-reads them into a live context.  dt_conv2d / dt_convlstm_step do the same on entry. */
DT_API int dt_policy_reload(dt_ctx *ctx);
/* One knob of ONE context without touching the process environment: name "pin", value 1 / 0 = DT_PIN for this context only
 * (kernel selection independent of the batch a call carries; object_tracking_amd/parallel.py: deterministic=True). */
DT_API int dt_policy_set(dt_ctx *ctx, const char *name, int value);

/* ---- profiling --------------------------------------------------------- */
/* When enabled every kernel launch is bracketed by HIP events on the ctx
 * stream.  dt_profile_read synchronises the stream and returns, per kernel
 * family, launches / total ms / algorithmic flops / algorithmic bytes since the
 * last reset.  names: "conv_igemm", "conv1_direct", "convlstm_gates",
 * "decode_nms", "associate", "lstm_step", "pool", "misc"; per-layer tags such as
 * "conv_igemm:conv_19", "conv_igemm:convlstm_step" are listed by dt_profile_names. */
DT_API int dt_profile_enable(dt_ctx *ctx, int on);
DT_API int dt_profile_reset(dt_ctx *ctx);
/* newline-separated list of every name with data (families and "family:layer" tags) */
DT_API int dt_profile_names(dt_ctx *ctx, char *buf, size_t buflen);
DT_API int dt_profile_read(dt_ctx *ctx, const char *name, int64_t *launches,
                    double *total_ms, double *flops, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* MI355_DT_H */
