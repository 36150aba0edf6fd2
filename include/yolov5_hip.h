/* yolov5_hip.h -- C-ABI of libyolov5_hip.so, the MI355X (gfx950) kernel library behind the YOLOv5 hot path.
 *
 * The reference (ultralytics/yolov5, pure Python) has no FFI of its own: its "operator API" for this path is the
 * set of Python callables named below.  Every entry point here is what a maintainer would bind (ctypes stub in
 * INTEGRATION.md) to replace the third-party native op those callables reach today.
 *
 * Conventions
 *   - plain pointers + sizes only; all device memory is owned by the caller (PyTorch caching allocator);
 *     the library never allocates per call (one 4 KiB zero page per process is created on first use).
 *   - every launch goes to the `stream` argument (a hipStream_t passed as void*); no internal synchronisation.
 *   - return 0 on success, <0 = y5_status; y5_last_error() gives the message of the last failure on this thread.
 *   - activations are NHWC ("channels last"): element (b,h,w,c) at ((b*H+h)*W+w)*ld + c, `ld` >= C is the
 *     pixel stride in elements so a tensor may be a channel slice of a wider buffer (concat-free writes).
 */
#ifndef YOLOV5_HIP_H
#define YOLOV5_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { Y5_OK = 0, Y5_ERR_BAD_ARG = -1, Y5_ERR_UNSUPPORTED = -2, Y5_ERR_RUNTIME = -3, Y5_ERR_WORKSPACE = -4 } y5_status;
typedef enum { Y5_F16 = 0, Y5_F32 = 1, Y5_U8 = 2 } y5_dtype;

int y5_version(void);                /* 10000*major + 100*minor + patch */
const char* y5_last_error(void);     /* thread-local message of the last non-zero return */

/* ---------------------------------------------------------------------------------------------------------
 * y5_conv2d_fwd -- fused convolution: y = act(conv(x, w) + bias) [+ residual], NHWC, MFMA implicit GEMM.
 * Replaces: models/common.py:90-92 `Conv.forward_fuse` (nn.Conv2d + SiLU after utils/torch_utils.py:224-254
 * BN folding), common.py:181 `Bottleneck` residual add, common.py:246/340/453 `torch.cat` (write into a channel
 * slice via ldy), yolov5s.yaml:36,41 `nn.Upsample(None,2,'nearest')` (y_up2), models/yolo.py:95 Detect conv.
 * w_packed: [Npad][Kpad] row-major, k = (kh, kw, c); Npad % 32 == 0, Kpad = K rounded up to 128 bytes, zero padded.
 * bias: fp32 [Npad].  residual/y_up2 may be NULL.  residual may alias y (in-place add).  y may be NULL when only
 * the upsampled copy y_up2 is wanted.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype;          /* Y5_F16 | Y5_F32 (storage type of x, w, y, residual; accumulation is always fp32) */
  int B, H, W;        /* input batch / height / width */
  int C1, ldx;        /* input channels, input pixel stride (elements); both multiples of 16 bytes */
  int OH, OW;         /* output height / width (checked against H,W,k,s,p) */
  int C2, ldy;        /* output channels, output pixel stride; both multiples of 16 bytes */
  int KH, KW, SH, SW, PH, PW;
  int act;            /* 0 identity, 1 SiLU */
  int Kpad, Npad;     /* packed filter dims */
  int ldr;            /* residual pixel stride (if residual != NULL) */
  int ld2;            /* y_up2 pixel stride (if y_up2 != NULL); y_up2 has spatial size 2*OH x 2*OW */
  int cfg;            /* workgroup tile configuration id (y5_conv_cfg_info), -1 = built-in heuristic */
  int max_blocks;     /* 0 = fill the GPU (CUs x occupancy) with persistent workgroups; >0 caps the grid (tests) */
  /* Optional output placement (all zero = dense): output pixel (b, oh, ow) is stored at (b, oh*out_mul_h + out_off_h,
   * ow*out_mul_w + out_off_w) of an out_H x out_W image with pixel stride ldy, and OH/OW are taken as given (taps that
   * fall outside the input read zeros).  Used by the data-gradient of strided convolutions (one launch per output
   * parity class); general implicit-GEMM configurations only. */
  int out_mul_h, out_mul_w, out_off_h, out_off_w, out_H, out_W;
  int split_n;  /* > 0: channels [0, split_n) are stored to y (pixel stride ldy), channels [split_n, C2) to the y_up2 ARGUMENT (pixel stride
                 * ld2, channel n - split_n) -- one GEMM for C3's cv1 + cv2 (models/common.py:246) whose halves land in different buffers;
                 * no upsampled replica, residual or output placement in that mode */
  int up_c, ld_up;  /* up_c > 0 (configurations 88 / 89, 1x1 s1 fp16 layers): `nn.Upsample(None, 2, 'nearest')` + `Concat` (models/yolov5s.yaml:36-37,
                 * 41-42; models/common.py:443-453) consumed VIRTUALLY -- input channels [0, up_c) of output pixel (b, oh, ow) are read from a
                 * LOW-resolution tensor (B, H/2, W/2, pixel stride ld_up) at (b, oh >> 1, ow >> 1), channels [up_c, C1) from x as usual (x = the
                 * concat buffer, whose first up_c channels are then never written: no 2x replica exists).  The low-resolution tensor is passed in
                 * the `residual` argument (such a layer has no residual); every other configuration rejects up_c > 0. */
} y5_conv_desc;

#define Y5_CONV_NUM_CFGS 97   /* 0..13 implicit-GEMM tiles (2 LDS stages), 14..21 streaming pointwise (1x1 s1, fp16),
                                22..29 implicit-GEMM tiles with a 3-stage LDS ring (fp16), 30..34 streaming 3x3 (small C, fp16),
                                35..39 256-row implicit-GEMM tiles with 2-4 stage rings, 4 or 8 waves (fp16, deep layers),
                                40..45 producer/consumer implicit GEMM: 4 MFMA waves + 4 LDS-DMA waves per workgroup (fp16),
                                46..49 high-occupancy 2-stage tiles (epilogue scratch inside the idle ring stage, fp16),
                                50..55 tiles 320 / 160 / 192 / 96 channels wide (yolov5x / yolov5m channel counts, fp16),
                                56 streaming pointwise 128 -> 256 channels, epilogue in two channel groups (P3 Detect head, fp16),
                                57..60 stream-K implicit GEMM (fp16; 128x128, 256x256, 256x128, 128x256 tiles, BK64): every workgroup multiplies
                                the same number of K chunks, tiles split across workgroups are combined through the registered workspace,
                                61..77 halo-resident 3x3 s1 or s2 (fp16, C1 % 32 == 0): a workgroup owns a spatial output tile, the input halo is
                                staged once per 32-channel chunk and serves all nine taps, only the filter streams per tap,
                                78..79 streaming 3x3 s1 64 -> 64 with the filter's MFMA fragments in registers (3 / 4 LDS stages),
                                80..83 streaming 3x3 with EIGHT waves per workgroup and one (two) stage(s) per wave: 64->64 s1, 32->64 s2, 32->32 s1,
                                84..87 streaming pointwise with eight waves per workgroup, one stage per wave: 128->128, 64->64, 128->64, 128->256,
                                88..89 implicit GEMM (128x128 tile: producer / consumer BK32 ring, plain 2-stage BK64) for 1x1 layers whose input is
                                Upsample(2) + Concat, read virtually from the low-resolution tensor (y5_conv_desc.up_c > 0 only),
                                90..92 halo-resident 3x3 with 64- / 128-pixel tiles (the stride-2 down-sampling layers: a 4 x 16 output tile reads a
                                9 x 33 input halo),
                                93..94 pointwise (1x1 s1, fp16, C1 % 32 == 0) with K streamed through an LDS ring and the whole 256- / 128-channel N tile
                                owned by one workgroup of eight waves (256 pixels): the deep 1x1 layers of P4 / P5,
                                95..96 implicit GEMM on the 256-row / 8-phase ping-pong structure (conv_g8.h; fp16, C1 % 8 == 0 and C1 >= 64 -- a K tile of a layer
                                with C1 % 64 != 0 spans two taps --, <= 32 taps, Npad <= 2048):
                                256 pixels x 256 channels per workgroup, K tile 64 in four half-tiles, two wave rows staggered by one barrier, counted
                                vmcnt(6) once per K tile -- the MFMA-bound 3x3 / deep 1x1 layers; 96: 256 pixels x 128 channels (three half-tiles per K
                                tile, three K tiles resident) */
/* Scratch for the stream-K configurations (57..60): `bytes` of device memory (256-byte aligned; bytes >= y5_conv_sk_workspace_bytes())
 * that the CALLER owns and keeps alive; registered per device, used by every later y5_conv2d_fwd with such a configuration on ANY
 * stream -- so launches that may overlap in time must not both use stream-K (the engine keeps it off its side-stream ops).  The
 * library zeroes it once here (asynchronously on `stream`).  ws = NULL unregisters.  Without it those configurations return
 * Y5_ERR_WORKSPACE (the autotuner skips them). */
size_t y5_conv_sk_workspace_bytes(void);
int y5_conv_set_sk_workspace(void* ws, size_t bytes, void* stream);
int y5_conv_num_cfgs(void);
/* Measurement aid: back-to-back independent v_mfma_f32_32x32x16_f16 on every SIMD (two waves each, no memory traffic) for `iters` iterations of
 * eight MFMAs per wave; reports the TFLOP/s the device sustains and the shader clock it held meanwhile (s_memtime against the 100 MHz
 * s_memrealtime).  scratch: >= 2 * CUs * 1024 + 64 bytes of device memory.  bench.py prints it beside the data-sheet peak. */
int y5_probe_mfma(void* scratch, size_t scratch_bytes, int iters, float* tflops, float* shader_ghz, void* stream);
/* CU budget of the persistent kernels (grids = budget x occupancy instead of CUs x occupancy).  utils/torch_utils.py:61-70 `smart_DDP` + train.py:404-410:
 * the reference overlaps the gradient all-reduce with backward; here the collective's kernels (RCCL, on the process group's stream) need workgroup slots
 * beside the backward plan's persistent workgroups, so yolov5_amd.torch_utils.HipDDP reserves r CUs (budget = CUs - r) while a process group exists.
 * n_cus = 0 restores the device's full count.  Process-global; takes effect at the next launch / plan build. */
int y5_set_cu_budget(int n_cus);
int y5_conv_cfg_info(int cfg, int* bm_pixels, int* bn_channels, int* k_bytes_per_stage);

int y5_conv2d_fwd(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                  const void* residual, void* y, void* y_up2, void* stream);
/* Same call timed with HIP events on `stream` (1 warm-up + `iters` launches); *ms = average per launch.
 * Used by the engine's per-layer tile autotuner and by bench.py. */
int y5_conv2d_time(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                   const void* residual, void* y, void* y_up2, int iters, void* stream, float* ms);

/* ---------------------------------------------------------------------------------------------------------
 * y5_conv_stem_fwd -- the backbone's first layer (models/yolov5s.yaml:17 `Conv [64, 6, 2, 2]`, common.py:74-92,
 * BN folded + SiLU) computed DIRECTLY from the caller's NCHW fp16 batch (train.py:379 / detect.py:206-210 input
 * contract): x (B, 3, H, W) fp16 contiguous -> y NHWC (B, H/2, W/2, C2) fp16 with pixel stride ldy.
 * w_stem: [Npad][144] fp16, k = (c*6 + kh)*8 + kw with the kw = 6,7 taps zero; bias fp32 [Npad]; Npad = 32 or 64.
 * Requires even H and W % 64 == 0 (else Y5_ERR_UNSUPPORTED: use y5_nchw_to_nhwc + y5_conv2d_fwd).
 * ------------------------------------------------------------------------------------------------------- */
int y5_conv_stem_fwd(const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias, int C2, int Npad,
                     void* y, int ldy, int max_blocks, void* stream);
/* The same launch WITHOUT bias and activation (train mode: BatchNorm with batch statistics follows as its own passes, models/common.py:86-88). */
int y5_conv_stem_fwd_raw(const void* x_nchw, int B, int H, int W, const void* w_stem, int C2, int Npad, void* y, int ldy, int max_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_nchw_to_nhwc -- input contract (train.py:379, val.py:259-262, detect.py:206-210, common.py:926):
 * NCHW {u8|f16|f32} -> NHWC {f16|f32} with channel padding to `ld` (pad channels written as 0) and a scale
 * (1/255 for u8 images, 1 for already-normalised floats).
 * ------------------------------------------------------------------------------------------------------- */
int y5_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C, int H, int W, int ld,
                    float scale, void* stream);

/* y5_nhwc_to_nchw -- export a NHWC slice as a contiguous NCHW tensor (API boundary: raw head outputs, Proto). */
int y5_nhwc_to_nchw(const void* src, int dtype, void* dst, int B, int C, int H, int W, int ld, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_sppf_pool -- models/common.py:331,338-340: y1 = m(x), y2 = m(y1), y3 = m(y2), m = MaxPool2d(k,1,k//2)
 * (-inf padding), written as channel slices 1..3 of the same NHWC buffer whose slice 0 holds x:
 * buf[..., 0:C] = x (input), buf[..., C:2C] = y1, [2C:3C] = y2, [3C:4C] = y3;  pixel stride ld >= 4C.
 * ------------------------------------------------------------------------------------------------------- */
int y5_sppf_pool(void* buf, int dtype, int B, int H, int W, int C, int ld, int k, void* stream);

/* y5_upsample2x -- nn.Upsample(None,2,'nearest') into a channel slice (only used when the producer conv could
 * not emit the replicated store itself). */
int y5_upsample2x(const void* src, int dtype, void* dst, int B, int H, int W, int C, int lds, int ldd, void* stream);

/* y5_copy_slice -- strided NHWC channel-slice copy (generic `Concat` fallback, common.py:453). */
int y5_copy_slice(const void* src, int dtype, void* dst, int npix, int C, int lds, int ldd, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_detect_decode -- models/yolo.py:96-115 (`Detect.forward`, eval branch) for one level:
 * logits NHWC (B,ny,nx, na*no) [pixel stride ld] ->
 *   z[b, row_off + a*ny*nx + iy*nx + ix, :]  (B, nrows_total, no)    xy=(2s-0.5+g)*stride, wh=(2s)^2*anchor_px,
 *                                              conf/cls = sigmoid; the last `nm` columns are copied raw (Segment)
 *   raw[b, a, iy, ix, :] (optional, may be NULL): the un-activated (bs,na,ny,nx,no) tensor the reference returns.
 * anchors_px: na*2 floats = anchors (grid units) * stride, i.e. the reference's anchor_grid.
 * ------------------------------------------------------------------------------------------------------- */
int y5_detect_decode(const void* logits, int dtype, int B, int ny, int nx, int na, int no, int nm, int ld,
                     float stride, const float* anchors_px, void* z, int z_dtype, long long nrows_total,
                     long long row_off, void* raw, void* stream);
/* ... additionally writing every row's objectness to obj_hint ((B, nrows_total), z's dtype; NULL: none) for y5_nms_batched_hint */
int y5_detect_decode_hint(const void* logits, int dtype, int B, int ny, int nx, int na, int no, int nm, int ld, float stride,
                          const float* anchors_px, void* z, int z_dtype, long long nrows_total, long long row_off, void* raw,
                          void* obj_hint, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_detect_head_fwd -- one pyramid level of `Detect.forward` in export / z-only mode (models/yolo.py:83-108 with
 * models/common.py:866): the 1x1 convolution `self.m[i]` (descriptor `d`: fp16, C1 = 128, 3 x 85 output channels stored as
 * Npad = 256, act = 0) and the decode of y5_detect_decode in one pass -- the logits are never written.  z rows as in
 * y5_detect_decode (fp16); ny*nx % 32 == 0, nrows_total / row_off / ny*nx multiples of 8.  Bit-identical to the two-call form.
 * Y5_ERR_UNSUPPORTED for any other shape (callers keep y5_conv2d_fwd + y5_detect_decode).  d->cfg == 87 selects the build with eight waves per
 * workgroup and one LDS stage per wave, any other value four waves with two stages (identical results; the engine times both).
 * ------------------------------------------------------------------------------------------------------- */
int y5_detect_head_fwd(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, int ny, int nx, float stride,
                       const float* anchors_px, void* z, long long nrows_total, long long row_off, void* stream);
int y5_detect_head_fwd_hint(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, int ny, int nx, float stride,
                            const float* anchors_px, void* z, long long nrows_total, long long row_off, void* obj_hint, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_nms_batched -- utils/general.py:658-767 `non_max_suppression` incl. torchvision.ops.nms (general.py:750).
 * pred: (bs, n, no) f16|f32, no = 5 + nc + nm.  All arithmetic is fp32 on the fp32 value of each element.
 * out:       (bs, max_det, 6+nm) fp32 rows [x1,y1,x2,y2,conf,cls,(mask..)] in descending-confidence order
 * out_count: (bs) int32 number of valid rows per image
 * flags: Y5_NMS_MULTI_LABEL | Y5_NMS_AGNOSTIC.  classes: optional device int32[nclasses] filter (general.py:734).
 * Tie rule: equal confidences keep the lower candidate index first (stable sort contract).
 * workspace: y5_nms_workspace_bytes(...) bytes, 256-byte aligned.
 * ------------------------------------------------------------------------------------------------------- */
enum { Y5_NMS_MULTI_LABEL = 1, Y5_NMS_AGNOSTIC = 2 };
size_t y5_nms_workspace_bytes(int bs, int n, int no, int nm, int flags, int max_nms);
int y5_nms_batched(const void* pred, int dtype, int bs, int n, int no, int nm, float conf_thres, float iou_thres,
                   int max_det, int max_nms, float max_wh, int flags, const int* classes, int nclasses,
                   float* out, int* out_count, void* workspace, size_t workspace_bytes, void* stream);
/* The same with an objectness HINT: obj_hint (bs, n), pred's dtype, holds (approximately) pred[..., 4] -- the plane y5_detect_decode_hint /
 * y5_detect_head_fwd_hint write beside z.  The filter reads 2 bytes per row instead of the whole row and fetches a row from `pred` only where
 * the plane cannot exclude it with a 2^-8 relative margin; every decision is then taken on `pred` itself, so results are those of y5_nms_batched.
 * obj_hint = NULL: identical to y5_nms_batched. */
int y5_nms_batched_hint(const void* pred, int dtype, int bs, int n, int no, int nm, float conf_thres, float iou_thres,
                        int max_det, int max_nms, float max_wh, int flags, const int* classes, int nclasses, float* out,
                        int* out_count, void* workspace, size_t workspace_bytes, const void* obj_hint, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_conv2d_wgrad -- weight gradient of the convolution described by `d` (same descriptor as the forward call; act,
 * ldy, ldr, ld2 ignored; max_blocks > 0 sets the number of pixel-range splits (the train engine times a few per layer
 * geometry and keeps the fastest; 0 = library default); k1 s1 p0 layers take a linear-staging build of the kernel).
 * d->cfg picks the kernel family: -1 / 0 = automatic, 1 = the general im2col-gather kernel (1xy: its filter tile capped at
 * 64 x output channels by 64 y k columns, x, y in {1,2} -- more tiles, fewer splits, less atomic traffic), 3 = the patch-staged kernel
 * of the 3x3 layers (k3 p1, stride 1 or 2: one workgroup owns all nine taps of a channel tile and stages spatial patches
 * -- the activation travels to LDS ~3 times instead of 9; wins where a filter element sees many pixels, P1-P3), 3xy = the
 * same with its channel tile capped at 32 x output channels by 32 y input channels (x in {1,2,4}, y in {1,2}), 6 = the stem
 * kernel (0.Conv on the paired-pixel view: k(6,3) s(2,1) p(2,1), C1 = 8, ldx = 8; the automatic choice for that geometry);
 * Y5_ERR_BAD_ARG if cfg >= 3 is asked for another geometry:
 *   dw_packed[n][k] += sum_pixels dz[pixel][n] * im2col(x)[pixel][k]     fp32, layout [Npad][Kpad] of w_packed.
 * The caller zero-fills dw_packed; accumulation uses fp32 atomics (split over the pixel range).  x, dz: fp16 NHWC
 * slices (pixel strides d->ldx, ld_dz).  Replaces autograd's conv weight backward under train.py:410.
 * ------------------------------------------------------------------------------------------------------- */
int y5_conv2d_wgrad(const y5_conv_desc* d, const void* x, const void* dz, int ld_dz, float* dw_packed, void* stream);
/* Deterministic form: the pixel-range splits write their partial sums into `workspace` ([splits][Npad][Kpad] fp32, size from
 * y5_conv2d_wgrad_ws_bytes for the same descriptor incl. max_blocks; 16-byte aligned) and a second launch adds them to dw_packed in split order --
 * bit-identical from run to run and from rank to rank, where the atomic form above is order-dependent in the last bits of fp32 (what
 * torch.use_deterministic_algorithms asks of cudnn's weight gradient in the reference's --seed runs, train.py:105 init_seeds(deterministic=True)). */
int y5_conv2d_wgrad_det(const y5_conv_desc* d, const void* x, const void* dz, int ld_dz, float* dw_packed, void* workspace, size_t workspace_bytes,
                        void* stream);
long long y5_conv2d_wgrad_ws_bytes(const y5_conv_desc* d, int ld_dz);   /* < 0: error (y5_last_error) */

/* ---------------------------------------------------------------------------------------------------------
 * Train-mode Conv block pieces (models/common.py:82-88 `Conv.forward` = SiLU(BatchNorm2d(conv(x))) with BATCH
 * statistics; eps / momentum as set by initialize_weights, models/yolo.py:259).  z = raw conv output (y5_conv2d_fwd
 * with act = 0 and zero bias), NHWC slice, npix = B*OH*OW pixels, C channels (C*elemsize % 16 == 0).
 * y5_bn_silu_fwd: batch mean / biased variance per channel (deterministic two-level reduction) -> save_mean,
 *   save_invstd = 1/sqrt(var+eps); running stats updated like torch (unbiased variance, momentum); then
 *   y = [residual +] silu(gamma*(z-mean)*invstd + beta)   (residual = Bottleneck shortcut, common.py:181).
 * y5_bn_silu_bwd: dy -> dz (gradient w.r.t. the conv output), dgamma, dbeta (fp32).
 * y5_channel_sum: out[c] = sum over pixels of x[., c]  (bias gradient of Detect.m[i], models/yolo.py:95).
 * workspace: y5_bn_workspace_bytes(C, npix) bytes, 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------- */
size_t y5_bn_workspace_bytes(int C, long long npix);
int y5_bn_silu_fwd(const void* z, int dtype, long long npix, int C, int ldz, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                   const void* residual, int ldr, void* y, int ldy, void* workspace, size_t workspace_bytes, void* stream);
int y5_bn_silu_bwd(const void* dy, int ld_dy, const void* z, int ldz, int dtype, long long npix, int C, const float* gamma,
                   const float* beta, const float* save_mean, const float* save_invstd, void* dz, int ld_dz, float* dgamma,
                   float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
int y5_channel_sum(const void* x, int dtype, long long npix, int C, int ld, float* out, void* workspace, size_t workspace_bytes,
                   void* stream);
/* SyncBatchNorm (train.py:269-271 `torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` under DDP): the two fused entries above cut at the point
 * where the ranks exchange per-channel sums -- the exchange (all-reduce SUM) is the host's, there is no collective at the C-ABI.
 *   forward :  y5_bn_stats (sums[0..C) = sum z, sums[C..2C) = sum z^2, fp64)  ->  all-reduce(sums), count_total = sum of the ranks' npix
 *              ->  y5_bn_silu_fwd_from_sums (mean / invstd / running statistics from the global sums, then the apply pass)
 *   backward:  y5_bn_bwd_stats (this rank's dgamma, dbeta: they ARE the parameter gradients, averaged later like every gradient)
 *              ->  all-reduce(copy of dgamma, dbeta)  ->  y5_bn_silu_bwd_from_sums (dz from the global sums and count_total)
 * With one rank (count_total = npix, sums untouched) the pairs compute what y5_bn_silu_fwd / y5_bn_silu_bwd compute. */
int y5_bn_stats(const void* z, int dtype, long long npix, int C, int ldz, double* sums, void* workspace, size_t workspace_bytes, void* stream);
/* Batch statistics from the EPILOGUE of the convolution that produces z (round 5; models/common.py:82-88 `Conv.forward` in train mode: the BatchNorm's
 * batch mean / variance are a function of the convolution's output alone).
 *   y5_conv2d_fwd_stats          = y5_conv2d_fwd (act = 0, no residual, one destination y = the z buffer) that also writes one row [2][C2] of fp32
 *                                  (sum z, sum z^2) per workgroup into `partial` -- sums of the fp16-ROUNDED outputs, added in a fixed order (deterministic);
 *                                  *rows = number of rows written (= grid size, <= 8 x CUs); partial_bytes >= rows * 2 * C2 * 4.
 *                                  Y5_ERR_UNSUPPORTED unless the chosen configuration is a streaming pointwise / 3x3 kernel (conv_pw.h / conv_k3.h: the
 *                                  HBM-bound layers of P1-P3), fp16: the caller then runs y5_conv2d_fwd + y5_bn_silu_fwd.
 *   y5_bn_silu_fwd_from_partials = y5_bn_silu_fwd without its statistics pass over z: fixed-order finish of the `rows` partial rows (mean / invstd / running
 *                                  statistics exactly as y5_bn_silu_fwd), then the apply pass. */
int y5_conv2d_fwd_stats(const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y, float* partial, size_t partial_bytes,
                        int* rows, void* stream);
int y5_bn_silu_fwd_from_partials(const void* z, int dtype, long long npix, int C, int ldz, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                 const float* partial, int rows, const void* residual, int ldr, void* y, int ldy, void* stream);
int y5_bn_silu_fwd_from_sums(const void* z, int dtype, long long npix, int C, int ldz, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, const double* sums,
                             long long count_total, const void* residual, int ldr, void* y, int ldy, void* stream);
int y5_bn_bwd_stats(const void* dy, int ld_dy, const void* z, int ldz, int dtype, long long npix, int C, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
int y5_bn_silu_bwd_from_sums(const void* dy, int ld_dy, const void* z, int ldz, int dtype, long long npix, int C, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_invstd, const float* sum_dgamma, const float* sum_dbeta,
                             long long count_total, void* dz, int ld_dz, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Training-path glue (fp16 NHWC slices; ld = pixel stride in elements, multiples of 8):
 * y5_nhwc_to_raw / y5_raw_to_nhwc -- models/yolo.py:96-98 `x[i].view(bs,na,no,ny,nx).permute(0,1,3,4,2)` and its
 *   backward: logits (B, npix, ld >= na*no) <-> raw (B, na, npix, no) (padding channels of dlogits are zeroed).
 * y5_upsample2x_bwd  -- backward of nn.Upsample(2,'nearest'): gsrc(b,h,w,:) (+)= sum of the 2x2 block of gup.
 * y5_add_slice       -- dst (+)= src over npix pixels x C channels (Bottleneck shortcut / Concat fan-out gradients).
 * y5_sppf_pool_bwd   -- backward of SPPF's three chained max-pools (common.py:338-340) in the [x|y1|y2|y3] buffers:
 *   on entry grad holds the per-slice gradients left by cv2's data-gradient, on return grad[..., 0:C] = d/dx.  Bit-reproducible: the scatter to the
 *   window maxima accumulates on an exact 2^-24 fixed-point grid (integer LDS atomics), k <= 15, H*W <= ~1700 (LDS).
 *   The gradients are fp16 (their magnitude is bounded by 65504: the grid holds every fp16 value exactly).  Non-finite incoming gradients (the fp16
 *   overflow a loss scaler must see): the fixed-point form (planes <= 1371 pixels) has no Inf / NaN on its integer grid and therefore writes NaN to the WHOLE
 *   output of the (image, channel-group) workgroup that met one; the gather form (larger planes, Y5_SPPF_BWD_GATHER) propagates Inf / NaN element by element
 *   as fp32 addition does.  Both leave a non-finite value in d/dx, which is all `GradScaler` asks for; code that inspects WHERE must not rely on either.
 *   Returns Y5_ERR_UNSUPPORTED when the plane does not fit the device's LDS.
 * ------------------------------------------------------------------------------------------------------- */
int y5_nhwc_to_raw(const void* logits, void* raw, int B, int npix, int na, int no, int ld, void* stream);
int y5_raw_to_nhwc(const void* draw, void* dlogits, int B, int npix, int na, int no, int ld, void* stream);
int y5_upsample2x_bwd(const void* gup, void* gsrc, int B, int H, int W, int C, int ld_up, int ld_src, int accumulate, void* stream);
int y5_add_slice(const void* src, void* dst, long long npix, int C, int lds, int ldd, int accumulate, void* stream);
int y5_sppf_pool_bwd(const void* act, void* grad, int B, int H, int W, int C, int ld_act, int ld_grad, int k, void* stream);
/* fp32 twins of the five glue calls above for the training plan's reference-precision mode (one entry, `op` selects):
 * 0 nhwc_to_raw (H = npix, W = na, C = no, a = ld)   1 raw_to_nhwc (same)   2 upsample2x_bwd (a = ld_up, b = ld_src, c = accumulate)
 * 3 add_slice (B*H*W pixels, a = lds, b = ldd, c = accumulate)   4 sppf_pool_bwd (src = activations, dst = gradients, a = ld_act, b = ld_grad, c = k).
 * y5_conv2d_wgrad and y5_filter_jobs (job.reserved = 1: fp32 destination) take fp32 through their existing signatures. */
int y5_train_glue_f32(int op, const void* src, void* dst, int B, int H, int W, int C, int a, int b, int c, void* stream);
/* Filter (re)packing on the device -- fp32 master weights (C2, C1, KH, KW) change at every optimizer step:
 * y5_pack_conv_weight : -> fp16 [Npad][Kpad], k = (kh*KW + kw)*C1_view + c (C1_view >= C1: channel padding of the stem view)
 * y5_pack_dgrad_weight: -> fp16 [Npad][Kpad] sub-filter of one data-gradient parity class, out[c1][(a*ntw + b)*C2_view + c2] =
 *                          w[c2][c1][taps_h[a]][taps_w[b]]  (taps_* are HOST arrays, at most 8 entries)
 * y5_unpack_conv_wgrad: fp32 [.][Kpad] weight gradient of y5_conv2d_wgrad -> (C2, C1, KH, KW) parameter layout
 * y5_memset_zero      : hipMemsetAsync on the stream (zero-fill of the weight-gradient accumulators) */
int y5_pack_conv_weight(const float* w, int C2, int C1, int KH, int KW, int C1_view, void* out_f16, int Kpad, int Npad, void* stream);
int y5_pack_dgrad_weight(const float* w, int C2, int C1, int KH, int KW, const int* taps_h, int nth, const int* taps_w, int ntw,
                         int C2_view, void* out_f16, int Kpad, int Npad, void* stream);
int y5_unpack_conv_wgrad(const float* dw_packed, int Kpad, float* gw, int C2, int C1, int KH, int KW, int C1_view, void* stream);
/* The same three transforms over many filters in ONE launch: `jobs_dev` is a DEVICE array of njobs descriptors, max_total the
 * largest `total` among them.  kind 0 = y5_pack_conv_weight, 1 = y5_pack_dgrad_weight, 2 = y5_unpack_conv_wgrad (total = output
 * elements: Npad*Kpad for the packs, C2*C1*KH*KW for the unpack), 3 = the stem filter (C2, 3, 6, 6) fp32 -> [Npad][144] fp16 of
 * y5_conv_stem_fwd / _raw (total = Npad * 144). */
typedef struct {
  const void* src; void* dst;
  long long total;
  int kind, C2, C1, KH, KW, C1_view, C2_view, Kpad, Npad, nth, ntw;
  int th[8], tw[8];
  int reserved;
} y5_filter_job;
int y5_filter_jobs(const y5_filter_job* jobs_dev, int njobs, long long max_total, void* stream);
int y5_memset_zero(void* p, size_t bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_loss_forward / y5_loss_backward -- utils/loss.py:101-247 `ComputeLoss.__call__` + `build_targets`, including
 * the un-vendored ultralytics `bbox_iou(CIoU=True)` / `smooth_bce` it calls (loss.py:6,117,153).
 * p[i]: device pointer of level i's raw head output (bs, na, ny[i], nx[i], 5+nc), contiguous, dtype `dtype`.
 * targets: device (nt, 6) fp32 rows [img, cls, x, y, w, h] (xywh normalised), nt may be 0.
 * forward:  out4 (device, 4 floats) = [loss = (lbox+lobj+lcls)*bs, lbox, lobj, lcls]          (loss.py:178-183)
 * backward: dp[i] = d(loss * *grad_scale)/dp[i], same shape and dtype as p[i]; grad_scale is a DEVICE scalar
 *           (autograd grad_output, e.g. GradScaler scale x WORLD_SIZE) or NULL for 1.  Needs the workspace of the
 *           matching y5_loss_forward call untouched.  No host synchronisation in either call.
 * Contracts: duplicate (b,a,gj,gi) rows -> tobj takes the LAST row's iou (loss.py:163 on CPU); row gradients of
 * duplicate cells are summed in ascending row order; gr = 1, autobalance off, sort_obj_iou off.
 * anchors: [level * 16 + a * 2 + {0,1}] in grid units (Detect.anchors); balance: loss.py:125.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct {
  int dtype;
  int nl, na, nc, bs;
  int ny[5], nx[5];
  float anchors[5 * 16];
  float balance[5];
  float hyp_box, hyp_obj, hyp_cls, cls_pw, obj_pw, anchor_t, cp, cn;
  float fl_gamma;  /* hyp['fl_gamma'] (loss.py:120-122): > 0 wraps both BCE terms in FocalLoss(gamma, alpha = 0.25), loss.py:77-98; 0 = plain BCE */
} y5_loss_desc;
size_t y5_loss_workspace_bytes(const y5_loss_desc* d, int nt);   /* 0 on invalid descriptor */
/* Byte offset, inside the workspace, of float obji[nl]: each level's mean objectness BCE before its balance factor (utils/loss.py:171),
 * valid after y5_loss_forward -- the input of ComputeLoss(autobalance=True) (utils/loss.py:173-177, host arithmetic).  -1 on invalid descriptor. */
long long y5_loss_obji_offset(const y5_loss_desc* d, int nt);
int y5_loss_forward(const y5_loss_desc* d, const void* const* p, const float* targets, int nt, float* out4,
                    void* workspace, size_t workspace_bytes, void* stream);
int y5_loss_backward(const y5_loss_desc* d, const void* const* p, int nt, const float* grad_scale, void* const* dp,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Byte offsets inside the workspace of level `level`'s build_targets result (loss.py:185-247), valid after
 * y5_loss_forward: offs = {n (int32), b, a, gj, gi, tcls (int32[cap]), tbox (float[cap][4]), anch (float[cap][2]),
 * iou (float[cap]), row_grad (float[cap][5+nc])}; *cap = 5*na*nt rows. */
int y5_loss_targets_layout(const y5_loss_desc* d, int nt, int level, size_t offs[10], long long* cap);

/* ---------------------------------------------------------------------------------------------------------
 * y5_process_mask -- utils/segment/general.py:25-51 `process_mask` (+ `crop_mask` :10-22) for ONE image:
 * protos (c, mh, mw) f16|f32 contiguous; instance i has c fp32 coefficients at masks_in + i*ld_m and an xyxy box (input
 * image pixels) at boxes + i*ld_b (both may point into the NMS output rows: ld = 6+nm).  out: (n, ih, iw) when
 * upsample != 0 (bilinear, align_corners=False, then > 0.5) else (n, mh, mw); element type Y5_F32 (0.f/1.f, the
 * reference's `masks.gt_(0.5)` result) or Y5_U8 (0/1).
 * ------------------------------------------------------------------------------------------------------- */
int y5_process_mask(const void* protos, int proto_dtype, int c, int mh, int mw, const float* masks_in, int ld_m,
                    const float* boxes, int ld_b, int n, int ih, int iw, int upsample, void* out, int out_dtype,
                    void* stream);
/* The same for a whole BATCH in one launch (segment/predict.py:161-172 calls process_mask per image): protos (B, c, mh, mw); image b has imgs[b].n instances,
 * coefficient rows at masks_in + i*ld_m and boxes at boxes + i*ld_b (device pointers, typically into the rows of the padded NMS output; `imgs` itself is a HOST
 * array, copied into the kernel arguments -- no device table, no host sync).  out: (sum of n, oh, ow), the instances of image 0 first; one grid over
 * (image, instance, 64-row x 256-byte tile) items, tiles outside the instance's box are plain zero stores.  Needs ow % 4 == 0 (Y5_F32) / ow % 16 == 0 (Y5_U8) and a
 * 16-byte aligned `out`; otherwise Y5_ERR_UNSUPPORTED (call y5_process_mask per image). */
typedef struct y5_mask_img { const float* masks_in; const float* boxes; int ld_m, ld_b, n; } y5_mask_img;
int y5_process_mask_batch(const void* protos, int proto_dtype, int B, int c, int mh, int mw, const y5_mask_img* imgs, int ih, int iw, int upsample,
                          void* out, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_letterbox_batch -- the image pre-processing chain for a batch in one launch:
 *   utils/augmentations.py:85-115 `letterbox` (cv2.resize INTER_LINEAR + cv2.copyMakeBorder with `pad_value`),
 *   utils/dataloaders.py LoadImages / detect.py:205 `im.transpose((2,0,1))[::-1]` (HWC->CHW, BGR->RGB when swap_rb),
 *   detect.py:208-209 / val.py:261-262 / models/common.py:926 `.half() / 255` (div255).
 * jobs_dev: device array of B jobs.  src: u8 HWC 3-channel image, h0 x w0 pixels, `stride` bytes per row; nw x nh is
 *   letterbox's `new_unpad` (== w0 x h0: no resize), (top, left) its border; the host mirror
 *   yolov5_amd/augmentations.py:letterbox_geometry computes them with the reference's rounding rules.
 * dst: (B, 3, H, W) when dst_chw != 0 else (B, H, W, 3); dtype Y5_U8 (the reference's letterbox output), Y5_F16 or Y5_F32.
 * The resize restates OpenCV's 8-bit INTER_LINEAR (11-bit fixed point, exact 2x down-scale -> 2x2 area mean); cv2 is a
 * third-party dependency of the reference that is absent from this image: that layer is parity-unpinned.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct y5_letterbox_job {
  const void* src;
  int h0, w0, stride;
  int nw, nh, top, left;
} y5_letterbox_job;
int y5_letterbox_batch(const y5_letterbox_job* jobs_dev, int B, int H, int W, int pad_value, int swap_rb, void* dst, int dst_dtype,
                       int dst_chw, int div255, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_val_match -- val.py:296-307 for every image of a batch in one launch: de-letterbox the predictions and the labels
 * (`scale_boxes` with ratio_pad, utils/general.py:613-626) and `process_batch` (utils/metrics.py:224-265, box branch;
 * `box_iou` of ultralytics.utils.metrics, call site utils/metrics.py:252).
 * det:       (bs, max_det, ld_det) fp32 rows [x1,y1,x2,y2,conf,cls,...] as y5_nms_batched writes them; det_count (bs)
 *            int32 valid rows per image (NULL: max_det rows everywhere); max_det <= 1024.
 * labels:    (nlabels, ld_lab) fp32 rows; column img_col holds the image index (img_col < 0: all rows belong to image 0),
 *            cls_col the class, box_col..box_col+3 the box: centre-x, centre-y, w, h when xywh != 0 (val.py:274 targets in
 *            letterboxed pixels), x1,y1,x2,y2 otherwise (process_batch's own labels layout).
 * scale:     optional (bs, 5) fp32 [gain, pad_x, pad_y, h0, w0] = shapes[si][1][0][0], shapes[si][1][1], shapes[si][0]
 *            (val.py:283,298); NULL: boxes are compared as given.
 * iouv:      (niou) fp32 thresholds, niou <= 32.
 * correct:   (bs, max_det, niou) uint8 0/1; rows past det_count are written as 0.
 * predn:     optional (bs, max_det, 4) fp32 native-space boxes (val.py:297-298) for save_json / save_txt.
 * Tie rule: a detection whose best IoU is shared by two labels of its class takes the later label row.
 * ------------------------------------------------------------------------------------------------------- */
int y5_val_match(const float* det, int ld_det, int max_det, const int* det_count, int bs, const float* labels, int ld_lab,
                 int nlabels, int img_col, int cls_col, int box_col, int xywh, const float* scale, const float* iouv, int niou,
                 unsigned char* correct, float* predn, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_scale_boxes_batch -- utils/general.py:613-626 `scale_boxes` (+ `clip_boxes` :629-640) applied IN PLACE to columns 0..3 of
 * the first det_count[i] rows of every image of the padded NMS output (call sites detect.py:248 with do_round,
 * models/common.py:941, val.py:298): x = clamp((x - pad) / gain, 0, w0|h0).  scale: (bs, 5) fp32 [gain, pad_x, pad_y, h0, w0].
 * ------------------------------------------------------------------------------------------------------- */
int y5_scale_boxes_batch(float* det, int ld_det, int max_det, const int* det_count, int bs, const float* scale, int do_round,
                         void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_conv_k3pw_fwd -- `Conv(c1, c2, 3, 2)` followed by a pointwise convolution as ONE launch (csrc/conv_k3.h, PW2): models/yolo.py:160-170
 * walking `1.Conv` into `2.C3`, whose cv1 and cv2 (common.py:246, one GEMM with N = 2 c_) are the only readers of the Conv's output -- the
 * 3x3's bias + SiLU result stays in LDS and is multiplied with the second filter by the wave that produced it.  d describes the 3x3 layer
 * (fp16, 3x3 s2 p1, C1 = 32, <= 64 output channels, SiLU; d->cfg = 31 / 34 or -1; OH % 4 == 0, OW % 8 == 0); the pointwise layer has C3 <= 64
 * output channels (filter packed [64][Kpad2], k = the 3x3's output channel, bias fp32 [64]), activation act2; its channels [0, split_n) go
 * to y (pixel stride ldy), [split_n, C3) to y2 (pixel stride ld2, channel n - split_n); split_n == C3: everything to y.
 * ------------------------------------------------------------------------------------------------------- */
int y5_conv_k3pw_fwd(const y5_conv_desc* d, const void* x, const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2,
                     int C3, int Npad2, int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_conv_front_fwd -- the first three layers of the backbone as ONE launch (csrc/conv_front.h): models/yolov5s.yaml:17-19 walked by
 * models/yolo.py:160-170 -- `0.Conv [64, 6, 2, 2]` read straight from the NCHW fp16 batch (train.py:379 / detect.py:206-210 input contract),
 * `1.Conv [128, 3, 2]`, and the merged `2.C3.cv1 + 2.C3.cv2` pointwise GEMM that is the only reader of 1.Conv (models/common.py:246), each
 * Conv.forward_fuse (common.py:90-92: folded-BN bias + SiLU).  The stem's and the 3x3's outputs stay in LDS / registers.
 * x (B, 3, H, W) fp16 contiguous, H % 64 == 0, W % 64 == 0.  w_stem / bias0 as y5_conv_stem_fwd (C0 = 32 output channels);
 * w1_packed [Npad1][Kpad1] fp16, k = (kh, kw, c) with c < 32 (y5_conv2d_fwd packing), bias1 fp32 [Npad1], C1 <= 64;
 * w2_packed [Npad2][Kpad2] fp16, k = the 3x3's output channel, bias2 fp32 [Npad2], C3 <= 64 in multiples of 16;
 * output channels [0, split_n) -> y (pixel stride ldy), [split_n, C3) -> y2 (pixel stride ld2, channel n - split_n); NHWC (B, H/4, W/4, .).
 * Y5_ERR_UNSUPPORTED for other shapes: use y5_conv_stem_fwd + y5_conv_k3pw_fwd.
 * ------------------------------------------------------------------------------------------------------- */
int y5_conv_front_fwd(const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias0, int C0, const void* w1_packed,
                      const float* bias1, int C1, int Npad1, int Kpad1, int act1, const void* w2_packed, const float* bias2, int C3, int Npad2,
                      int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n, int max_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_sppf_cv1_pool_fwd -- models/common.py:318-340 `SPPF.forward` up to the concat: x1 = cv1(x) (1x1 C1 -> c_, BN folded, bias + SiLU), y1 = m(x1),
 * y2 = m(y1), y3 = m(y2) with m = MaxPool2d(k, 1, k // 2), as ONE launch (csrc/conv_sppf.h): a workgroup owns all H x W pixels of one image for 64
 * output channels, so the pools run on the LDS copy of what the GEMM just produced.  The four results land in channel slices [s c_, (s + 1) c_),
 * s = 0..3, of `buf` (NHWC, pixel stride ld >= 4 c_): the concat of :340 is never materialised separately -- cv2 reads `buf`.
 * fp16; H * W <= 416 (P5 of a 640 x 640 input: 20 x 20), C1 % 32 == 0, c_ % 64 == 0, odd k; w_packed [c_][Kpad] (k = c) as y5_conv2d_fwd.
 * Y5_ERR_UNSUPPORTED otherwise: y5_conv2d_fwd + y5_sppf_pool.
 * ------------------------------------------------------------------------------------------------------- */
int y5_sppf_cv1_pool_fwd(const void* x, int ldx, const void* w_packed, const float* bias, int Kpad, void* buf, int ld, int B, int H, int W, int C1,
                         int c_, int k, int act, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_bottleneck_fwd -- models/common.py:164-181 `Bottleneck.forward` inside C3 (e = 1.0, :242): y = [x +] cv2(cv1(x)) with cv1 = 1x1
 * C->C and cv2 = 3x3 pad 1 C->C (BN folded, bias + SiLU each), fp16, as ONE pass: the 1x1 output stays in LDS (csrc/conv_bneck.h).
 * x / y: NHWC channel slices with pixel strides ldx / ldy (elements); y must NOT overlap x.  C = 32 or 64 (H % 4 == 0, W % 8 == 0), or
 * C = 128 (csrc/conv_h3b.h: GEMM-1 phase in front of the halo-resident 3x3; any H, W; max_blocks bits 16.. select the ring form, 0 = default).
 * Filters packed like y5_conv2d_fwd's ([32-padded C][Kpad], k = (kh, kw, c)), biases fp32 [C].
 * ------------------------------------------------------------------------------------------------------- */
int y5_bottleneck_fwd(const void* x, int ldx, const void* w1_packed, const float* bias1, int Kpad1, const void* w2_packed, const float* bias2,
                      int Kpad2, void* y, int ldy, int B, int H, int W, int C, int add, int max_blocks, void* stream);
/* The last Bottleneck of a C3 + the C3's cv3 (models/common.py:246: cv3(cat(m(cv1(x)), cv2(x)))) as ONE launch: the Bottleneck's result
 * stays in LDS and is the first half of the 1x1's input, y2 (C3's cv2 output: NHWC slice, pixel stride ld2) the second; out (pixel stride
 * ldo, C3 <= 2 C channels) = act3(W3 [y ; y2] + b3) with W3 packed [2 C padded][Kpad3], k = (y's C channels, then y2's).  C = 32 (csrc/conv_bneck.h)
 * or C = 128 (csrc/conv_h3b.h CV3 form: cv3 as a GEMM-3 phase on the LDS-resident result; measured slower than two launches). */
int y5_bottleneck_cv3_fwd(const void* x, int ldx, const void* w1_packed, const float* bias1, int Kpad1, const void* w2_packed, const float* bias2,
                          int Kpad2, const void* y2, int ld2, const void* w3_packed, const float* bias3, int Kpad3, int C3, int act3, void* out,
                          int ldo, int B, int H, int W, int C, int add, int max_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Test-time augmentation glue (models/yolo.py:269-312 `_forward_augment`; the three forwards in between are ordinary plan runs).
 * y5_scale_img   -- utils/torch_utils.py `scale_img(img.flip(f), ratio, gs)` of an NCHW batch: optional flip of the source (2 = up-down,
 *   3 = left-right), F.interpolate(size = (OH, OW), bilinear, align_corners = False), F.pad to (PH, PW) with pad_value (0.447) at the
 *   right / bottom.  src: u8 (read as x / 255) | f16 | f32; dst: (B, C, PH, PW) f16 | f32.  fp32 weights as torch; tolerance 1e-6 (fp32).
 * y5_tta_descale -- `_descale_pred` in place on (rows, no) predictions: [..., :4] /= scale; flip 2: y = img_h - y; flip 3: x = img_w - x.
 * ------------------------------------------------------------------------------------------------------- */
int y5_scale_img(const void* src, int src_dtype, int B, int C, int H, int W, int flip, int OH, int OW, int PH, int PW, float pad_value,
                 void* dst, int dst_dtype, void* stream);
int y5_tta_descale(void* z, int dtype, long long rows, int no, float scale, int flip, float img_h, float img_w, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * y5_mosaic_batch -- the training input pipeline for a whole batch in one launch: utils/dataloaders.py:798-855 `load_mosaic`
 * (four images resized to the training size by `load_image` :770-790 and tiled on a 2s x 2s canvas of 114s; or, where the hyp['mosaic']
 * gate of :701 sends a sample down the letterbox branch :710-733, that one image on an s x s canvas -- job.canvas), the image half of
 * utils/augmentations.py:118-166 `random_perspective` (cv2.warpAffine, INTER_LINEAR, border 114, output s x s), :69-83 `augment_hsv`,
 * the flips of dataloaders.py:747-757, `img.transpose((2, 0, 1))[::-1]` (:761) and collate_fn's torch.stack (:862).  Draws, geometry
 * and labels are the host's (yolov5_amd/dataloaders.py); jobs_dev is a DEVICE array of B descriptors (+ the mixup partners behind them).  Source images: uint8 HWC BGR.
 * dst: (B, 3, S, S) RGB planes, uint8 / fp16 / fp32 (div255: divide by 255 like train.py:375).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct {
  const void* src[4];            /* the four source images of the mosaic in tile order (NULL = no tile) */
  int h0[4], w0[4], stride[4];   /* original size, row stride in bytes */
  int rh[4], rw[4];              /* size after load_image's resize (longest side = s) */
  int x1a[4], y1a[4], x2a[4], y2a[4];  /* tile rectangle on the canvas (dataloaders.py:812-822) */
  int x1b[4], y1b[4];            /* top-left corner of the part of the resized image that lands there */
  double A[6];                   /* INVERSE affine map of cv2.warpAffine: src = A @ (x, y, 1) */
  unsigned char lut[3][256];     /* hue / saturation / value look-up tables of augment_hsv */
  int hsv, flipud, fliplr;
  int canvas;                    /* side of the square canvas the tiles sit on; 0 = 2 S (mosaic).  S for the non-mosaic branch of
                                    dataloaders.py:710-733: load_image + letterbox(auto=False) = ONE tile at (left, top) of an S x S canvas */
  double mix_r;                  /* mixup (utils/augmentations.py:225-233, dataloaders.py:707-708): out = uint8(this * mix_r + partner * (1 - mix_r)) */
  int mix_job;                   /* 1 + index of the partner job in the table (partners sit behind the B rendered jobs), 0 = no mixup; of the partner
                                    only the tiles, rectangles, A and canvas are used */
  int reserved;
} y5_mosaic_job;
int y5_mosaic_batch(const y5_mosaic_job* jobs_dev, int B, int S, int pad_value, void* dst, int dst_dtype, int div255, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Execution plan: a recorded list of the calls above, replayed by ONE host call (and optionally through a
 * captured hipGraph).  Replaces the Python module walk of models/yolo.py:160-170 `_forward_once`.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct y5_plan y5_plan;
y5_plan* y5_plan_create(void);
void y5_plan_destroy(y5_plan*);
int y5_plan_add_conv(y5_plan*, const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                     const void* residual, void* y, void* y_up2);
int y5_plan_add_detect_head(y5_plan*, const y5_conv_desc* d, const void* x, const void* w_packed, const float* bias, int ny, int nx,
                            float stride, const float* anchors_px, void* z, long long nrows_total, long long row_off);
int y5_plan_add_bottleneck(y5_plan*, const void* x, int ldx, const void* w1_packed, const float* bias1, int Kpad1, const void* w2_packed,
                           const float* bias2, int Kpad2, void* y, int ldy, int B, int H, int W, int C, int add);
int y5_plan_add_conv_k3pw(y5_plan*, const y5_conv_desc* d, const void* x, const void* w1_packed, const float* bias1, const void* w2_packed,
                          const float* bias2, int C3, int Npad2, int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n);
int y5_plan_add_conv_front(y5_plan*, const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias0, int C0, const void* w1_packed,
                           const float* bias1, int C1, int Npad1, int Kpad1, int act1, const void* w2_packed, const float* bias2, int C3, int Npad2,
                           int Kpad2, int act2, void* y, int ldy, void* y2, int ld2, int split_n);  /* x_nchw may be NULL: y5_plan_set_input */
int y5_plan_set_obj_hint(y5_plan*, int op_index, void* obj_hint);  /* Detect decode / fused head op: also write the objectness plane */
int y5_plan_add_bottleneck_cv3(y5_plan*, const void* x, int ldx, const void* w1_packed, const float* bias1, int Kpad1, const void* w2_packed,
                               const float* bias2, int Kpad2, const void* y2, int ld2, const void* w3_packed, const float* bias3, int Kpad3, int C3,
                               int act3, void* out, int ldo, int B, int H, int W, int C, int add);
int y5_plan_add_nop(y5_plan*);  /* placeholder op: keeps the op numbering of the conv + decode form next to a fused head */
int y5_plan_add_conv_stem(y5_plan*, const void* x_nchw, int B, int H, int W, const void* w_stem, const float* bias, int C2,
                          int Npad, void* y, int ldy);
int y5_plan_set_input(y5_plan*, int op_index, const void* src);  /* re-point a stem / nchw_to_nhwc op at a new input batch */
/* Side branch: consecutive ops marked branch = 1 run on a plan-owned second stream, forked behind the main op that precedes them in the
 * list and joined at the end of the executed range (in a captured graph: fork / join edges).  The caller guarantees that their inputs are
 * complete at the fork point and that no later op of the range reads their outputs (the Detect heads of the lower pyramid levels,
 * models/yolo.py:83-108, which only the final output depends on). */
int y5_plan_set_branch(y5_plan*, int op_index, int branch);
/* Tile configuration of a recorded y5_plan_add_conv op (the host's in-situ refinement of the tuner's isolated race: engine.py _refine_in_situ times the
 * runner-up of every layer inside the running plan).  Only before the first graph capture; Y5_ERR_BAD_ARG for any other op kind. */
int y5_plan_set_conv_cfg(y5_plan*, int op_index, int cfg);
int y5_plan_set_anchors(y5_plan*, int op_index, const float* anchors_px, int n);  /* decode / fused-head op: new anchor sizes (px) */
int y5_plan_add_nchw_to_nhwc(y5_plan*, const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C, int H,
                             int W, int ld, float scale);
int y5_plan_add_sppf_pool(y5_plan*, void* buf, int dtype, int B, int H, int W, int C, int ld, int k);
int y5_plan_add_sppf_cv1_pool(y5_plan* plan, const void* x, int ldx, const void* w_packed, const float* bias, int Kpad, void* buf, int ld, int B, int H, int W,
                              int C1, int c_, int k, int act);
int y5_plan_add_upsample2x(y5_plan*, const void* src, int dtype, void* dst, int B, int H, int W, int C, int lds, int ldd);
int y5_plan_add_copy_slice(y5_plan*, const void* src, int dtype, void* dst, int npix, int C, int lds, int ldd);
int y5_plan_add_detect_decode(y5_plan*, const void* logits, int dtype, int B, int ny, int nx, int na, int no, int nm,
                              int ld, float stride, const float* anchors_px, void* z, int z_dtype,
                              long long nrows_total, long long row_off, void* raw);
int y5_plan_add_nhwc_to_nchw(y5_plan*, const void* src, int dtype, void* dst, int B, int C, int H, int W, int ld);
int y5_plan_size(const y5_plan*);
int y5_plan_run(y5_plan*, void* stream);                 /* eager replay of every recorded op */
int y5_plan_run_range(y5_plan*, int first, int last, void* stream);
int y5_plan_capture(y5_plan*, void* stream);             /* capture the replay into a hipGraph (once) */
int y5_plan_capture_range(y5_plan*, int first, int last, void* stream);  /* same for ops [first, last) */
int y5_plan_launch_graph(y5_plan*, void* stream);        /* hipGraphLaunch of the captured graph */

/* Timing helper for bench/profiling: run ops [first,last) `iters` times on `stream` bracketed by hipEvents
 * recorded on that same stream; returns total milliseconds in *ms. */
int y5_plan_time_range(y5_plan*, int first, int last, int iters, void* stream, float* ms);
/* In-situ per-op timing: ops [first,last) run once per pass in plan order on `stream` with a hipEvent between consecutive ops
 * (every op sees the inputs / cache state its predecessor left); ms_per_op[k - first] = median over `iters` passes.  Side-branch
 * ops run on `stream` too.  Synchronises. */
int y5_plan_profile_range(y5_plan*, int first, int last, int iters, void* stream, float* ms_per_op);
/* Fresh outputs per call (the reference returns new tensors, models/yolo.py:115): re-point every op output of [first,last) that equals
 * old_ptr at new_ptr; y5_plan_select_graph(key) picks the graph captured under that binding key (1 = found, 0 = capture needed: the
 * next y5_plan_capture_range is stored under `key`; up to 8 graphs per plan, least recently selected evicted). */
int y5_plan_rebind_output(y5_plan*, int first, int last, const void* old_ptr, void* new_ptr);
int y5_plan_select_graph(y5_plan*, unsigned long long key);

/* ---------------------------------------------------------------------------------------------------------
 * Fused optimizer step over all parameter tensors -- train.py:413-421 `scaler.unscale_(optimizer)`,
 * `clip_grad_norm_(model.parameters(), max_norm=10.0)`, `scaler.step(optimizer)` with the SGD(momentum, nesterov) groups of
 * utils/torch_utils.py:257-290 `smart_optimizer`, and `ema.update(model)` (utils/torch_utils.py:354-365).
 * The caller keeps a DEVICE array of y5_mt_tensor rows; every pointer is fp32 device memory of n elements.
 *   y5_mt_grad_norm : stats[0] = || grad * inv_scale ||_2 over all tensors (deterministic order), stats[1] = clip coefficient
 *                     min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0), stats[2] = 1.0 if any gradient is inf/nan
 *   y5_mt_sgd_step  : d = grad * inv_scale * stats[1]; d += wd[group] * p; buf = momentum * buf + d (buffers start at zero, so
 *                     the first step gives buf = d like torch's clone); d = nesterov ? d + momentum * buf : buf; p -= lr[group] * d;
 *                     if (ema) ema = ema_decay * ema + (1 - ema_decay) * p.  stats == NULL: no clipping / no skip;
 *                     stats[2] != 0: parameters and momentum are left untouched (GradScaler skips the step), the EMA still moves.
 *                     mom == NULL for a tensor: plain SGD on it.
 *   y5_mt_lerp      : ema = decay * ema + (1 - decay) * param over a table (ModelEMA over float buffers).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
  void* param;       /* fp32 [n], updated in place                    */
  const void* grad;  /* fp32 [n]                                      */
  void* mom;         /* fp32 [n] momentum buffer or NULL              */
  void* ema;         /* fp32 [n] exponential moving average or NULL   */
  long long n;
  int group;         /* 0..3: index into lr4 / wd4                    */
  int reserved;
} y5_mt_tensor;
size_t y5_mt_workspace_bytes(int ntensors, long long max_numel);
int y5_mt_grad_norm(const y5_mt_tensor* table_dev, int ntensors, long long max_numel, float inv_scale, float max_norm,
                    float* stats_dev /* 4 floats */, void* workspace, size_t workspace_bytes, void* stream);
int y5_mt_sgd_step(const y5_mt_tensor* table_dev, int ntensors, long long max_numel, const float* lr4, const float* wd4, float momentum,
                   int nesterov, float inv_scale, const float* stats_dev, float ema_decay, void* stream);
int y5_mt_lerp(const y5_mt_tensor* table_dev, int ntensors, long long max_numel, float decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLOV5_HIP_H */
