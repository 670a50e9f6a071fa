/*
 * clx.h — C ABI of libclx.so, the MI355X (gfx950) kernel library behind
 * cellulus_amd.  Plain C: caller-owned DEVICE pointers, explicit extents, a HIP
 * stream handle.  No allocation and no ownership transfer inside the library.
 * Every entry point is asynchronous on `stream` and returns 0 on success or a
 * negative clx_status; clx_last_error() returns the message of the last
 * failure on the calling thread.
 *
 * The reference (funkelab/cellulus) is pure Python and has no FFI; what each
 * entry point replaces is the *library call* the reference makes on the hot
 * path.  The replaced call site is cited per function as
 * `cellulus/<file>:<line>` (paths relative to the reference checkout).
 *
 * Tensor layout used by all kernels: channels-last ("pixel-major"):
 *   element (b, z, y, x, c) of a (B, D, H, W) grid with C channels sits at
 *   ptr[(((b*D + z)*H + y)*W + x) * ld + c],  ld >= C, ld % 4 == 0, C % 4 == 0
 * (2-D data uses D == 1).  The reference's NCHW tensors are converted at the
 * model boundary with clx_planar_to_pixel / clx_pixel_to_planar.
 */
#ifndef CLX_H
#define CLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* clx_stream; /* hipStream_t */

enum clx_status {
  CLX_OK = 0,
  CLX_ERR_ARG = -1,    /* invalid argument / unsupported shape */
  CLX_ERR_LAUNCH = -2, /* HIP launch or runtime failure */
  CLX_ERR_WORKSPACE = -3
};

const char* clx_last_error(void);
/* Library/ABI version (bumped when a signature changes). */
int clx_abi_version(void);
/* Number of HIP devices visible to the library (0 = none; not an error). */
int clx_device_count(void);

/* Kernel timing for roofline reports: while enabled (on = 1; on = 2 also clears the
 * records, 0 disables and clears) every launch of an MFMA kernel — and of the HBM-bound kernels of
 * detect / segment listed below — carries a HIP event pair stamped with the kernel's own start and end on its stream.  clx_profile_read sums launches / milliseconds / executed FLOPs
 * (2*M*N*K per GEMM of the launch, real extents) of one kernel kind; it synchronises. */
enum clx_profile_kind {
  CLX_PROF_IGEMM_WIDE = 0,   /* conv_igemm_kernel<128,128> */
  CLX_PROF_IGEMM_NARROW = 1, /* conv_igemm_kernel<128,64>  */
  CLX_PROF_WGRAD = 2,        /* conv_wgrad_kernel<...>     */
  CLX_PROF_GEMM_SP = 3,      /* gemm_sp_kernel: split-precision products (clx_conv_desc.precision = CLX_PREC_F32X3BF16;
                                FLOPs = the f32-equivalent 2*M*N*K) */
  CLX_PROF_WGRAD_SP = 4,     /* wgrad_sp_kernel: the weight gradient of the same precision */
  CLX_PROF_SPLIT_PLANES = 5, /* sp_split_kernel (HBM-bound: no FLOPs) */
  CLX_PROF_CHAIN64 = 6,      /* chain64_fwd / _bwd kernels (fused pairs of 64-channel 1x1 layers) */
  /* the HBM-bound kernels of detect / segment (no FLOPs: total_flops of these kinds is 0; the caller prices them by bytes) */
  CLX_PROF_MS_PREPARE = 7,   /* ms_prepare_kernel */
  CLX_PROF_MS_ASSIGN = 8,    /* ms_assign_cells / _grid / ms_assign kernels */
  CLX_PROF_CC = 9,           /* every kernel of clx_label_filter (connected components + size filter + relabel) */
  CLX_PROF_GROW_SHRINK = 10, /* every kernel of clx_grow_shrink */
  CLX_PROF_MINMAX = 11,      /* minmax_init + minmax_kernel */
  CLX_PROF_HISTOGRAM = 12,   /* histogram_kernel */
  CLX_PROF_NOISE_STATS = 13, /* noise_stats kernels */
  CLX_PROF_WINO_FUSED = 14,  /* wino_fused_kernel (CLX_ALGO_WINOGRAD4_FUSED; FLOPs = 2 * a^2 * tiles * N * C executed) */
  CLX_PROF_WINO_TRANSFORM = 15, /* the HBM-bound transform kernels of CLX_ALGO_WINOGRAD / _WINOGRAD4 (no FLOPs) */
  CLX_PROF_GEMM_SP2 = 16     /* gemm_sp2_kernel: the split-precision product on 128 x 128 tiles, two workgroups per CU (K <= 1024) */
};
int clx_profile_enable(int on);
int clx_profile_read(int kind, double* launches, double* total_ms, double* total_flops);
/* Shader clock the MFMA kernels actually ran at: the middle block of every clx_conv_fwd GEMM launch adds the shader-clock
 * ticks and the 100-MHz wall-clock ticks of its own life to two device counters; this reads (and optionally resets)
 * them: MHz = 100 * shader_ticks / wall_ticks.  Synchronises the device.  (Measurement only: no reference counterpart.) */
int clx_profile_clock(double* shader_ticks, double* wall_ticks_100mhz, int reset);

/* ------------------------------------------------------------------------ */
/* Convolution (valid, stride 1, kernel extent 1 or 3 per dim)              */
/* replaces: nn.Conv{2,3}d + nn.ReLU inside funlib ConvPass and the 1x1     */
/* head (cellulus/models/unet.py:24-63), their autograd backward            */
/* (cellulus/train.py:178), nn.Upsample(nearest) + centre-crop + torch.cat  */
/* of the U-Net's right path (fused into the A-operand gather).             */
/* ------------------------------------------------------------------------ */

/* One input source of a convolution.  The convolution reads a *logical* input
 * grid of extent (ID, IH, IW); logical voxel l of source s is stored at grid
 * position ((l + o) / f) of the source's (D, H, W) array — `o` is a crop
 * offset, `f` a nearest-neighbour upsampling factor (1 = none). */
typedef struct clx_src {
  const float* ptr;
  int C;          /* channels taken from this source (multiple of 4) */
  int ld;         /* floats between consecutive pixels */
  int D, H, W;    /* stored grid extent */
  int oz, oy, ox; /* crop offset (in logical = upsampled coordinates) */
  int fz, fy, fx; /* nearest upsample factor per dim, >= 1 */
} clx_src;

typedef struct clx_conv_desc {
  int nsrc;       /* 1 or 2; channels of src[0] come first (torch.cat order) */
  clx_src src[2];
  int B;
  int ID, IH, IW; /* logical input extent */
  int KD, KH, KW; /* kernel extent, each 1, 2 or 3 */
  int PD, PH, PW; /* zero padding per side (0 for the valid forward conv) */
  int N;          /* output channels actually computed (<= ld_out) */
  const float* wpack; /* [N][KD*KH*KW][Ctot] packed weights (clx_pack_weights) */
  const float* bias;  /* [N] or NULL */
  int relu;           /* epilogue: max(0, .) */
  const float* mask;  /* optional [M][ld_mask]: out *= (mask > 0) (ReLU backward) */
  int ld_mask;
  float* out;         /* [M][ld_out], M = B*OD*OH*OW, O = I + 2P - K + 1 */
  int ld_out;
  int accumulate;     /* epilogue: out = act(conv + bias + out) (residual already in `out`) */
  int algo;           /* clx_conv_algo: 0 = direct implicit GEMM */
  void* workspace;    /* CLX_ALGO_WINOGRAD*: clx_conv_workspace_bytes() bytes of scratch */
  size_t workspace_bytes;
  /* Winograd only, optional: a^2 * T * C floats (the head of the workspace layout) that hold the
   * transformed input V = B^T d B instead of the workspace.  clx_conv_fwd writes it; a later
   * clx_conv_wgrad of the same layer with vcache_valid = 1 reads it and skips its own input
   * transform (the forward and the weight gradient transform the same tensor); clx_conv_fwd with
   * vcache_valid = 1 takes V from it instead of transforming (see dy_vcache). */
  void* vcache;
  int vcache_valid;
  /* Hint, 0 = unknown: only the first c_real channels of src[0] can be non-zero (the rest is the
   * padding of a 1-, 2- or 3-channel raw image up to 4).  Kernels may skip the padding; the weight
   * gradient then leaves the padded channels of dwpack untouched (clx_unpack_wgrad drops them). */
  int c_real;
  /* Winograd, optional: clx_conv_wgrad also writes the input transform of dY that the data gradient
   * of the same layer needs (zero padding K - 1) into this buffer — a^2 * T_d * N floats, T_d = B *
   * output planes * ceil((OH + K - 1) / tile) * ceil((OW + K - 1) / tile) — reading dY once for both
   * transforms; the following clx_conv_fwd (data-gradient form) then sets vcache = this buffer and
   * vcache_valid = 1 and skips its own input transform.  dY must be dense (ld_dy == N). */
  void* dy_vcache;
  /* ReLU gates as bits (optional, both may be NULL).  gate_out: with relu = 1, also write
   * bit (n & 31) of word gate_out[m * ld_gate + (n >> 5)] = (out[m][n] > 0); requires ld_out % 32 == 0
   * (whole words per pixel; bits of channels >= N are written as 0).  mask_bits: the same layout
   * read INSTEAD of `mask` in the epilogue (out *= gate): 1/32 of the bytes of the float mask, which
   * is what the data-gradient of a 1x1 layer — HBM-bound at 64 channels — otherwise reads in full. */
  unsigned int* gate_out;
  int ld_gate;
  const unsigned int* mask_bits;
  int ld_mask_bits;
  /* clx_conv_wgrad only, optional: run-to-run REPRODUCIBLE weight gradients (the reference's CPU
   * autograd, cellulus/train.py:177-179, is deterministic; the default split-K combine is not:
   * float atomics arrive in any order).  With det_turns != NULL (clx_conv_wgrad_turns_bytes(d)
   * bytes of device scratch) the pixel slices of an output tile add their partial sums in slice
   * order — each block waits for its predecessor's turn counter — and the first-layer kernels
   * (whose block sums meet in LDS atomics) are not used.  Pass dbias = NULL with it and take the
   * bias gradient from clx_colsum_ordered. */
  int* det_turns;
  /* clx_conv_fwd in its data-gradient form, CLX_ALGO_WINOGRAD4 with a 3x3 (2-D) or 3x3x3 kernel only: ADJOINT
   * form.  The call must directly follow the clx_conv_wgrad of the same layer with the same workspace: the
   * weight gradient's A dY A^T (left in the workspace) is the operand — dX = sum over tiles of
   * B [U^T (A dY A^T)] B^T — so dY is not transformed a second time (src[0].ptr is not read).  wpack must come
   * from clx_pack_weights(CLX_PACK_WINO4_ADJOINT).  Epilogues: mask / mask_bits (no bias, ReLU, accumulate). */
  int adjoint;
  /* clx_conv_fwd, Winograd algorithms on 2-D layers (KD = 1) only, optional: ALSO write the 2 x 2 max-pooled output
   * (funlib's Downsample after a level's last convolution, cellulus/models/unet.py:24-51: nn.MaxPool2d(2) of this very
   * tensor) — pool_out[(b, y / 2, x / 2)][n] = max over the 2 x 2 window of out, after bias / ReLU / mask —, whose four
   * pixels lie inside one output tile of the transform: the separate pooling pass (a full read of `out`) disappears.
   * Output height and width must be even; ld_pool % 4 == 0; the same values clx_maxpool_fwd(out, 1, 2, 2) writes. */
  float* pool_out;
  int ld_pool;
  /* clx_conv_fwd, Winograd algorithms on 2-D layers (KD = 1) only, optional: compute ONLY the output tiles listed —
   * tile_list[i] = (b * th + ty) * tw + tx with th x tw = ceil(OH / tile) x ceil(OW / tile) tiles per image — and leave
   * every other element of `out` (and `pool_out`) untouched: the noisy copies of one image differ from it in some
   * tiles only (the caller has put the clean image's output under every copy).  The transforms and the
   * batched products see tile_count tiles; each listed tile gets the bits the full computation gives it.  No
   * vcache / accumulate with a list. */
  const int* tile_list;
  int tile_count;
  /* clx_conv_precision.  CLX_PREC_F32 (0, default): float32 MFMA — the reference's arithmetic.  CLX_PREC_F32X3BF16: where the
   * convolution is a plain matrix product — 1x1 layers, the transform-domain products of the 2-D Winograd layers,
   * forward, data gradient and weight gradient; N % 128 == 0, contraction length % 64 == 0 and >= 128 — the operands
   * are split exactly into three bfloat16 pieces and six exact products per float32 product are accumulated in float32
   * on the bf16 matrix cores ("P3 planes" above; csrc/gemm_sp.hip).  Everything else stays on the float32 kernels.
   * The planes of the weights come from the caller (wplanes: clx_split_planes of `wpack` seen as [rows][K], K = the
   * product's contraction length — Ctot for a 1x1 layer, KD * C per transform point for a Winograd layer — refreshed
   * whenever wpack is); those of the activations are made by the library: by the Winograd transforms themselves inside
   * `workspace` / `vcache` (clx_conv_workspace_bytes and clx_conv_vcache_bytes grow accordingly), by a split pass into
   * `aplanes` (clx_planes_bytes(M, C) bytes of scratch) for a 1x1 layer.  A layer without wplanes (or a 1x1 layer
   * without aplanes) runs in float32.
   * aplanes_valid = 1: `aplanes` already holds the planes of src[0] (an earlier call of the same layer left them: the
   * forward pass for the weight gradient, the weight gradient's dyplanes for the data gradient) — no split pass.
   * dyplanes (clx_conv_wgrad of a 1x1 layer only): clx_planes_bytes(M, N) bytes that receive the planes of dY; the bias
   * gradient comes out of that split pass.
   * out_planes / out_colsum (clx_conv_fwd of a 1x1 layer that runs in this precision, dense output ld_out == N): the
   * epilogue ALSO writes the P3 planes of `out` (clx_planes_bytes(M, N) bytes: the operand planes of the layer that reads
   * `out` next — its aplanes with aplanes_valid = 1, or a weight gradient's dyplanes with dyplanes_valid = 1) and adds the
   * column sums of `out` into out_colsum[N] (the bias gradient of the layer whose dY this data-gradient call produces).
   * clx_conv_sp_covers(d) says whether a call will honour them. */
  int precision;
  const void* wplanes;
  void* aplanes;
  int aplanes_valid;
  void* dyplanes;
  int dyplanes_valid;   /* clx_conv_wgrad: dyplanes already hold the planes of dY (and dbias has been taken care of) */
  void* out_planes;
  float* out_colsum;
} clx_conv_desc;

enum clx_conv_precision { CLX_PREC_F32 = 0, CLX_PREC_F32X3BF16 = 1 };

enum clx_conv_algo {
  CLX_ALGO_DIRECT = 0,
  /* Winograd F(2x2, 3x3) for 2-D 3x3 convolutions (KD = 1, one source, no upsampling):
   * input transform -> 16 batched f32-MFMA GEMMs over C -> output transform; 2.25x fewer
   * multiplications than the direct form, f32 throughout (error ~7e-7 of the output range on a
   * 768-channel layer, the direct kernel ~4e-7).
   * wpack must come from clx_pack_weights(CLX_PACK_WINO_FWD / _WINO_DGRAD). */
  CLX_ALGO_WINOGRAD = 1,
  /* Winograd F(4x4, 3x3), interpolation points {0, 1, -1, 1/2, -2}: 36 batched GEMMs per 4x4
   * outputs, 4x fewer multiplications than the direct form; f32 throughout, error ~5e-6 of the
   * output range on a 768-channel layer (F(2x2): ~7e-7, direct: ~4e-7).
   * wpack must come from clx_pack_weights(CLX_PACK_WINO4_FWD / _WINO4_DGRAD).
   * Also accepts 2x2 kernels (F(4x4, 2x2), points {0, 1, -1, 1/2}: 25 GEMMs per 4x4 outputs
   * instead of 64 multiplications) — the low-resolution half of the sub-pixel upsample conv —
   * and 3-D layers with cubic kernels (3x3x3, 2x2x2; PD = PH): the transform is applied in
   * (y, x) per z plane and the z taps stay a contraction inside the batched GEMMs
   * (K = KD * C), i.e. 4x / 2.56x fewer multiplications in 3-D as well. */
  CLX_ALGO_WINOGRAD4 = 2,
  /* The same F(4x4, 3x3) / F(4x4, 2x2) arithmetic for 2-D valid layers in ONE launch and without workspace: a
   * workgroup owns 32 output tiles x 64 output channels and ALL 36 (25) transform-domain products of them — the input
   * transform of 8 channels at a time goes through LDS into the MFMA A operand, the 36 x 32 x 64 accumulators live
   * in the registers of 8 waves, the weights arrive as ready-made B fragments (CLX_PACK_WINO4_FUSED), the output
   * transform + bias / ReLU / accumulate / gate bits / 2 x 2 pooling runs on the accumulators through LDS.  The
   * transformed tensors V and M (2.25x the activation each, written and read once by CLX_ALGO_WINOGRAD4) never
   * exist in HBM.  Layers with N > 64 given clx_conv_fused_workspace_bytes(d) of workspace run as TWO launches instead
   * (input transform once, then products + output transform: only M stays on chip).  Requires
   * clx_conv_fused_applicable(d); honours tile_list / pool_out / gate_out / accumulate, knows neither mask / mask_bits
   * nor vcache (forward form only). */
  CLX_ALGO_WINOGRAD4_FUSED = 3
};
/* 1 if clx_conv_fwd accepts `d` with algo = CLX_ALGO_WINOGRAD4_FUSED (2-D valid 3x3 or 2x2 layer, one plain source with
 * C % 8 == 0 and a tensor below 4 GB, N % 64 == 0), else 0. */
int clx_conv_fused_applicable(const clx_conv_desc* d);
/* Scratch bytes CLX_ALGO_WINOGRAD4_FUSED wants in clx_conv_desc.workspace (0 = none: N = 64, everything in one launch).
 * With it (N > 64) the input is transformed ONCE by a launch of its own into the workspace, in the MFMA-fragment order
 * the product kernel loads straight into registers, and only the products' results M stay on chip; without it the
 * one-launch form runs (every block of 64 output channels transforms its input again). */
size_t clx_conv_fused_workspace_bytes(const clx_conv_desc* d);
enum clx_conv_pass { CLX_PASS_FWD = 0, CLX_PASS_WGRAD = 1 };
/* Scratch bytes clx_conv_fwd (pass FWD; also the dgrad form) / clx_conv_wgrad (pass WGRAD)
 * need for descriptor `d` with algo = CLX_ALGO_WINOGRAD / _WINOGRAD4; 0 if Winograd does not apply to
 * the geometry (the caller must then use CLX_ALGO_DIRECT). */
size_t clx_conv_workspace_bytes(const clx_conv_desc* d, int pass);
/* Bytes of the buffers a caller may hand to a Winograd layer to carry transformed tensors from one call to the next:
 * which = 0: clx_conv_desc.vcache (V = B^T d B of the layer's input: written by clx_conv_fwd, reused by clx_conv_wgrad);
 * which = 1: clx_conv_desc.dy_vcache (the data gradient's input transform of dY, written by clx_conv_wgrad).  Float32
 * tensors, or P3 planes — 1.5x the bytes — where the layer runs in the split precision.  0 if Winograd does not apply. */
size_t clx_conv_vcache_bytes(const clx_conv_desc* d, int which);

/* ---- Split-precision operands ("P3 planes", round 6) ----
 * The precision CLX_PREC_F32X3BF16 of clx_conv_desc computes float32 products on the bf16 matrix cores: every operand
 * element is split EXACTLY into three bfloat16 pieces x = h0 + h1 + h2 (truncation, 8 + 8 + 8 significand bits) and
 * the six products a_i b_j, i + j <= 2, are accumulated in float32 (v_mfma_f32_32x32x16_bf16) — the dropped terms are
 * <= 2^-24 relative, one float32 rounding.  The pieces are made ONCE where a tensor is produced, in the layout the
 * matrix core consumes ("P3"): an [R][K] operand (K % 16 == 0) as 1-KB fragments, fragment (rb, ks, p) = piece p of
 * rows 32 rb .. 32 rb + 31, k = 16 ks .. 16 ks + 15 at byte ((rb * K/16 + ks) * 3 + p) * 1024; inside a fragment the 16 bytes
 * at 512 h + 16 r hold x_p[32 rb + r][16 ks + 8 h .. + 7].  Rows up to the next multiple of 64 (at least 128 rows) exist and are ZERO.  6 bytes
 * per element.  (No reference counterpart: the reference's torch.nn.Conv{2,3}d keep float32 operands,
 * cellulus/models/unet.py:24-63.) */
/* bytes of the P3 planes of an [rows][K] operand (0 if K % 16 != 0) */
size_t clx_planes_bytes(long long rows, int K);
/* planes <- split(x[rows][ld], columns [0, K)).  x and planes 16-byte aligned, ld % 4 == 0, K % 16 == 0. */
int clx_split_planes(const float* x, long long ld, long long rows, int K, void* planes, clx_stream stream);
/* x[rows][ld] <- h0 + h1 + h2 of the planes (exact; tests and diagnostics) */
int clx_join_planes(const void* planes, long long rows, int K, float* x, long long ld, clx_stream stream);
/* out[m][n] = act(sum_k A[m][k] B[n][k] + bias[n]) from the P3 planes of A ([M][K]) and B ([N][K]): N % 128 == 0,
 * K % 64 == 0, ld_out % 4 == 0.  The plain-product form of clx_conv_fwd with precision = CLX_PREC_F32X3BF16 (which
 * splits / reuses planes by itself); exported for tests and for callers that keep planes of their own. */
int clx_gemm_planes(const void* a_planes, const void* b_planes, int M, int N, int K, const float* bias, int relu,
                    float* out, int ld_out, clx_stream stream);
/* dw[n][c] += sum over rows of dY[row][n] * x[row][c] from the P3 planes of dY ([rows][N]) and x ([rows][C]), N % 128 == 0,
 * C % 128 == 0, rows >= 128: float atomics into the caller's (zeroed or accumulating) dw[N][ld_dw].  The plain-product form
 * of clx_conv_wgrad with precision = CLX_PREC_F32X3BF16 (replaces the autograd weight gradient of nn.Conv{2,3}d,
 * cellulus/train.py:178). */
int clx_wgrad_planes(const void* dy_planes, const void* x_planes, long long rows, int N, int C, float* dw, int ld_dw,
                     clx_stream stream);

/* 1 if clx_conv_fwd(d) runs as the split-precision product from planes (1x1 layer, precision / wplanes / aplanes set, N % 128
 * == 0, C % 64 == 0, C >= 128) — the form that honours out_planes / out_colsum —, else 0 */
int clx_conv_sp_covers(const clx_conv_desc* d);

/* out = act(conv(in) + bias).  f32 MFMA implicit GEMM (M = output pixels,
 * N = output channels, K = taps x channels). Also used for the data gradient
 * (PD = K-1, weights packed with CLX_PACK_DGRAD). */
int clx_conv_fwd(const clx_conv_desc* d, clx_stream stream);

/* Bytes of device scratch clx_conv_desc.det_turns needs for this layer (0 if d is NULL). */
size_t clx_conv_wgrad_turns_bytes(const clx_conv_desc* d);
/* out[n] += sum_m x[m][n] (ADDS into out, like the bias path of clx_conv_wgrad: zero it once per step) in a FIXED order (block partials over contiguous row ranges, then one pass
 * over the partials in block order): the reproducible bias gradient, db = column sums of dY
 * (autograd of nn.ConvNd's bias, cellulus/train.py:178).  scratch: clx_colsum_scratch_bytes(N). */
size_t clx_colsum_scratch_bytes(int N);
int clx_colsum_ordered(const float* x, int ld, long long M, int N, float* out, void* scratch, clx_stream stream);

/* Two consecutive 1x1 convolutions over 64 channels in one pass over the pixels — the
 * `conv_pass.2 -> conv_pass.4` pair of every funlib ConvPass the reference builds
 * (cellulus/models/unet.py:24-51: kernel sizes [3, 1, 1, 3]) and the head
 * `head.0 -> head.2` (unet.py:52-63), where the level is 64 channels wide (HBM-bound layers):
 *   y1 = relu(x w1^T + b1)           [M][ld_y1]  (y1 may be NULL: inference keeps nothing)
 *   y2 = act(y1 w2^T + b2)           [M][ld_y2], N2 = 64 or <= 32 channels, act = ReLU if relu2
 * w1 / w2: forward packs of clx_pack_weights (CLX_PACK_FWD, one tap): [64][64] and [pad4(N2)][64].
 * gate1 / gate2: optional ReLU gates as bits (layout of clx_conv_desc.gate_out).
 * Same arithmetic as two clx_conv_fwd calls (f32 MFMA, f32 accumulation over the 64 channels). */
int clx_chain64_fwd(const float* x, int ld_x, long long M, const float* w1, const float* b1, float* y1,
                    int ld_y1, unsigned int* gate1, int ld_gate1, const float* w2, const float* b2, int N2,
                    int relu2, float* y2, int ld_y2, unsigned int* gate2, int ld_gate2, clx_stream stream);
/* Backward of that pair in one pass (autograd of the two convolutions in cellulus/train.py:178):
 *   dP1 = (dp2 w2) * (y1 > 0)   — never written —   dp0 = (dP1 w1) * (x > 0 if gate_x)
 *   dw2[n][c] += sum_p dp2[p][n] y1[p][c],  db2[n] += sum_p dp2[p][n]      ([pad4(N2)][64], [N2])
 *   dw1[n][c] += sum_p dP1[p][n] x[p][c],   db1[n] += sum_p dP1[p][n]      ([64][64], [64])
 * dp2: gradient w.r.t. layer 2's PRE-activation, [M][ld_dp2], N2 = 64 or <= 8 channels;
 * w2t / w1t: data-gradient packs (CLX_PACK_DGRAD, one tap): [64][pad4(N2)] and [64][64];
 * dw*: the dwpack layout of clx_conv_wgrad (one tap), accumulated with float atomics;
 * dp0 may be NULL (no data gradient wanted), db* may be NULL. */
int clx_chain64_bwd(const float* dp2, int ld_dp2, int N2, const float* y1, int ld_y1, const float* x,
                    int ld_x, int gate_x, long long M, const float* w2t, const float* w1t, float* dp0,
                    int ld_dp0, float* dw2, float* db2, float* dw1, float* db1, clx_stream stream);

/* Weight gradient of the convolution described by `d` (d->out/bias/relu/mask/
 * wpack ignored): dwpack[tap][n][c] += sum_p dy[p][n] * in[p (+) tap][c],
 * dbias[n] += sum_p dy[p][n]  (dbias may be NULL).  Both outputs are
 * ACCUMULATED with float atomics (split-K): zero them first.
 * dy: [M][ld_dy] gradient w.r.t. the pre-activation output. */
int clx_conv_wgrad(const clx_conv_desc* d, const float* dy, int ld_dy,
                   float* dwpack, float* dbias, clx_stream stream);

enum clx_pack_mode {
  CLX_PACK_FWD = 0,        /* w[n][c][tap] -> wp[n][tap][cpad]               */
  CLX_PACK_DGRAD = 1,      /* w[n][c][tap] -> wp[c][flip(tap)][npad] (rows c < cpad) */
  CLX_PACK_WINO_FWD = 2,   /* 3x3 only: U[16][cout_pad][cin_pad] = G g G^T             */
  CLX_PACK_WINO_DGRAD = 3, /* 3x3 only: U[16][cin_pad][cout_pad] of the flipped filter */
  CLX_PACK_WINO4_FWD = 4,  /* F(4x4): taps 9 -> U[36][cout_pad][cin_pad], taps 4 (2x2) -> U[25][..],
                            * taps 27 / 8 (3-D) -> U[36 | 25][cout_pad][kd][cin_pad]            */
  CLX_PACK_WINO4_DGRAD = 5, /* the same for the flipped filter: U[a*a][cin_pad][kd][cout_pad]    */
  CLX_PACK_WINO4_ADJOINT = 6, /* F(4x4, 3x3[x3]): the FORWARD filter transform stored transposed (z taps reversed),
                               * U[36][cin_pad][kd][cout_pad] (clx_conv_desc.adjoint) */
  CLX_PACK_WINO4_FUSED = 7    /* 2-D F(4x4, 3x3) / F(4x4, 2x2) for CLX_ALGO_WINOGRAD4_FUSED: the values of _WINO4_FWD in the
                               * MFMA-fragment order the fused kernel's waves load straight into registers,
                               * [cout_pad / 64][cin_pad / 8][36 | 25][2][64 lanes][4]; cout_pad % 64 == 0, cin_pad % 8 == 0 */
};
/* Repack torch-layout conv weights w (Cout, Cin, taps) for clx_conv_fwd.
 * cin_pad/cout_pad >= real extents (multiples of 4), padding is zero-filled.
 * FWD:   wp is [Cout][taps][cin_pad].
 * DGRAD: wp is [cin_pad][taps][cout_pad] with the taps reversed. */
int clx_pack_weights(const float* w, float* wp, int cout, int cin, int taps,
                     int cin_pad, int cout_pad, int mode, clx_stream stream);
/* All the weight packings of a step in ONE launch (a step packs ~20 small layers, forward and data-gradient
 * form each: as separate launches they are latency-bound).  `jobs`: DEVICE array of njobs descriptors, each
 * exactly the arguments of one clx_pack_weights call; max_total: the largest element count of a job's packed
 * output (sizes the grid).  Same results as the single calls. */
typedef struct clx_pack_job {
  const float* w;
  float* wp;
  int cout, cin, taps, cin_pad, cout_pad, mode;
} clx_pack_job;
int clx_pack_weights_batch(const clx_pack_job* jobs, int njobs, long long max_total, clx_stream stream);
/* dw[n][c][tap] = dwpack[tap][n][c] for n < cout, c < cin  (wgrad output ->
 * torch layout). dwpack is [taps][rows][cin_pad], rows >= cout. */
int clx_unpack_wgrad(const float* dwpack, float* dw, int cout, int cin, int taps,
                     int rows, int cin_pad, clx_stream stream);
/* Winograd wgrad output dU[a*a][rows][cin_pad] (a = tile + ksize - 1; (tile, ksize) = (2, 3),
 * (4, 3) or (4, 2)) -> dw[n][c][kd][ksize x ksize] = G^T dU G per z tap (torch layout);
 * kd = 1 for 2-D layers, kd = ksize for the 3-D form of CLX_ALGO_WINOGRAD4. */
int clx_unpack_wgrad_wino(const float* du, float* dw, int cout, int cin, int rows,
                          int cin_pad, int tile, int ksize, int kd, clx_stream stream);

/* (B, C, n) planar <-> (B, n, ld) pixel-major; channels c >= C of the
 * pixel-major side are written as zero / ignored. */
int clx_planar_to_pixel(const float* planar, float* pixel, int B, int C,
                        long long n, int ld, clx_stream stream);
int clx_pixel_to_planar(const float* pixel, float* planar, int B, int C,
                        long long n, int ld, clx_stream stream);

/* Sub-pixel re-indexing used to run the convolution over a nearest-upsampled tensor on
 * the LOW-resolution grid: with P = fz*fy*fx phases,
 *   depth_to_space: hi[(b, z*fz+a, y*fy+bb, x*fx+c)][n] = lo[(b,z,y,x)][((a*fy+bb)*fx+c)*N + n]
 *   space_to_depth: the inverse gather.
 * lo: (B, D, H, W) grid, P*N channels (pixel stride ld_lo); hi: (B, D*fz, H*fy, W*fx) grid,
 * N channels (pixel stride ld_hi). N % 4 == 0. */
int clx_depth_to_space(const float* lo, int ld_lo, float* hi, int ld_hi, int B, int D, int H,
                       int W, int N, int fz, int fy, int fx, clx_stream stream);
int clx_space_to_depth(const float* hi, int ld_hi, float* lo, int ld_lo, int B, int D, int H,
                       int W, int N, int fz, int fy, int fx, clx_stream stream);

/* Weights of the sub-pixel form: w (cout, cin, kd, kh, kw) with cin = C0 (skip half) + C1
 * (upsampled half) -> w_skip (cout, C0, taps) and weff (P*N, C1, ztaps), P = fz*fy*fx phases, N >=
 * cout rows per phase (padding rows zero), low-res kernel 2 along axes with factor 2 (taps {0,1} |
 * {2} for even, {0} | {1,2} for odd output positions summed), unchanged along axes with factor 1.
 * clx_subpixel_fold_grads is the adjoint: gradients of w_skip / weff -> gradient of w. */
int clx_subpixel_split_weights(const float* w, float* w_skip, float* weff, int cout, int cin,
                               int C0, int N, int kd, int kh, int kw, int fz, int fy, int fx,
                               clx_stream stream);
int clx_subpixel_fold_grads(const float* g_skip, const float* g_eff, float* gw, int cout, int cin,
                            int C0, int N, int kd, int kh, int kw, int fz, int fy, int fx,
                            clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Max pooling / upsample backward (funlib Downsample / Upsample,           */
/* cellulus/models/unet.py:24-51)                                           */
/* ------------------------------------------------------------------------ */
/* y = maxpool(x), window = stride = (fz, fy, fx); extents must divide. */
int clx_maxpool_fwd(const float* x, float* y, int B, int D, int H, int W, int C,
                    int fz, int fy, int fx, clx_stream stream);
/* Gradient w.r.t. the PRE-activation of the tensor x that feeds both the
 * max-pool and the cropped skip connection:
 *   g[p][c]  = (x[p][c] is the first maximum of its window ? dy_pool[win][c] : 0)
 *            + (p inside the crop ? dskip[p - crop][c] : 0)
 *   dx[p][c] = g * (x[p][c] > 0)
 * dskip may be NULL. dskip has extent (SD, SH, SW), pixel stride ld_skip and
 * sits at offset (cz, cy, cx) inside x's grid. */
int clx_maxpool_bwd(const float* x, const float* y, const float* dy_pool,
                    const float* dskip, int ld_skip, int SD, int SH, int SW,
                    int cz, int cy, int cx, float* dx, int B, int D, int H,
                    int W, int C, int fz, int fy, int fx, clx_stream stream);
/* Backward of nearest upsample (+ crop) into the PRE-activation gradient of the
 * low-resolution tensor y (extent D,H,W; C channels):
 *   dy[q][c] = (sum over the f-block of dcat[(q*f + r) - o][coff + c]) * (y > 0)
 * dcat: gradient of the concatenated tensor, logical extent (LD, LH, LW),
 * pixel stride ld_cat, channel offset coff; o = crop offset. */
int clx_upsample_bwd(const float* dcat, int ld_cat, int coff, int LD, int LH,
                     int LW, int oz, int oy, int ox, const float* y, float* dy,
                     int B, int D, int H, int W, int C, int fz, int fy, int fx,
                     clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Embedding gather, OCE loss, Adam                                         */
/* ------------------------------------------------------------------------ */
/* sel[b][p][c] = offsets[b][c][coord...] + coord[b][p][c]
 * replaces UNetModel.select_and_add_coordinates (cellulus/models/unet.py:108-124).
 * offsets: planar (B, ND, [Z,] Y, X) f32; coords: (B, P, ND) int64, column 0
 * indexes the LAST spatial axis.  Index rules of the reference's advanced indexing: -n..-1
 * wrap around, anything else outside [0, n) is an IndexError there; here such a row is never
 * dereferenced, its selection is NaN and *oob_count (device int, zeroed by the caller, may be
 * NULL) counts it so that the host can raise. */
int clx_gather_add_fwd(const float* offsets, const long long* coords, float* sel,
                       int B, int P, int ND, int Z, int Y, int X, int* oob_count,
                       clx_stream stream);
/* doffsets[b][c][coord] += dsel[b][p][c]  (float atomics; zero doffsets first); out-of-range
 * rows are skipped and counted as above */
int clx_gather_add_bwd(const float* dsel, const long long* coords, float* doffsets,
                       int B, int P, int ND, int Z, int Y, int X, int* oob_count,
                       clx_stream stream);
/* OCE loss forward + gradient, replaces OCELoss.forward
 * (cellulus/criterions/oce_loss.py:45-63) and its autograd backward:
 *   d = |a - r|, oce = sum(1 - exp(-d^2/T)), reg = w * sum |a|
 *   da = (2/T) exp(-d^2/T) (a - r) + w a/|a|      (0 where the norm is 0)
 * a, r: (npairs, ND) f32. sums[0..2] += (loss, oce, reg) in f64 (zero first).
 * da may be NULL (forward only); grad_scale multiplies da. */
int clx_oce_loss_fwd_bwd(const float* a, const float* r, float* da, double* sums,
                         long long npairs, int ND, float temperature,
                         float reg_weight, float grad_scale, clx_stream stream);
/* Fused train-step tail: gather(anchor), gather(reference), OCE loss, and the
 * scatter-add of the anchor gradient straight into doffsets (planar, zeroed by
 * the caller). Equivalent to the three calls above composed as in
 * cellulus/train.py:170-178.  sums: FOUR doubles, zeroed by the caller: (loss, oce, reg) and
 * the number of pairs with an out-of-range coordinate (skipped, never dereferenced; the
 * reference raises IndexError there, so must the caller when sums[3] != 0). */
int clx_oce_pairs_fused(const float* offsets, const long long* anchor,
                        const long long* reference, float* doffsets, double* sums,
                        int B, int P, int ND, int Z, int Y, int X,
                        float temperature, float reg_weight, clx_stream stream);
/* The same with REPRODUCIBLE results: the scatter-add of the anchor gradients (every anchor pixel occurs
 * ~31 times) accumulates 2^-40 fixed-point integers — integer addition is associative, so the sum
 * does not depend on the order the atomics arrive in — and the loss sums are block partials reduced in
 * block order.  `sums` is ADDED to, as by clx_oce_pairs_fused; doffsets is overwritten.
 * scratch: clx_oce_pairs_det_scratch_bytes(B, ND, Z * Y * X), need not be zeroed. */
size_t clx_oce_pairs_det_scratch_bytes(int B, int ND, long long npix);
int clx_oce_pairs_fused_det(const float* offsets, const long long* anchor, const long long* reference,
                            float* doffsets, double* sums, int B, int P, int ND, int Z, int Y, int X,
                            float temperature, float reg_weight, void* scratch, clx_stream stream);

/* Optional on-device pair sampler with the distribution of ZarrDataset.sample_coordinates
 * (cellulus/datasets/zarr_dataset.py:177-242) — NOT its random stream: anchor column d uniform on
 * the integers [lo, hi[d]] (hi: HOST array of ND ints), each anchor repeated num_refs times,
 * reference = anchor + a uniformly chosen row of `offsets` (DEVICE, noffsets x ND int32: the
 * admissible offsets |o|^2 < kappa^2, o != 0).  anchor / reference: (B, num_anchors*num_refs, ND)
 * int64 out.  Counter-based: the same (seed, stream_id) gives the same pairs. */
int clx_sample_pairs(long long* anchor, long long* reference, const int* offsets, int noffsets,
                     int B, int num_anchors, int num_refs, int ND, int lo, const int* hi,
                     unsigned long long seed, unsigned long long stream_id, clx_stream stream);
/* torch.optim.Adam(lr, betas, eps, weight_decay) single step with coupled L2
 * (cellulus/train.py:80-82,179) over flat buffers of n floats.
 * step = 1-based step count AFTER increment. */
int clx_adam_step(float* param, const float* grad, float* exp_avg,
                  float* exp_avg_sq, long long n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step,
                  clx_stream stream);
/* The same step, enqueued BEFORE the host has read the loss kernel's bad-coordinate count: the kernel does
 * nothing when the device double *skip_if_positive is > 0 (NULL = unconditional).  cellulus/train.py:171-179
 * raises IndexError from the coordinate indexing before optimizer.step(); here the step is already in the
 * stream when the host looks, and is a no-op exactly in that case. */
int clx_adam_step_guarded(float* param, const float* grad, float* exp_avg,
                          float* exp_avg_sq, long long n, double lr, double beta1,
                          double beta2, double eps, double weight_decay, int step,
                          const double* skip_if_positive, clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Inference statistics (cellulus/models/unet.py:90-98)                     */
/* ------------------------------------------------------------------------ */
/* preds: (T, C, n) planar predictions of T noisy forwards of one sample.
 * out: (C+1, n): out[c] = mean_t preds[t][c], out[C] = sum_c std_t (population
 * std, torch.std_mean(unbiased=False)). */
int clx_noise_stats(const float* preds, float* out, int T, int C, long long n,
                    clx_stream stream);
/* The same, ALSO keeping the running minimum and maximum of the std plane out[C] in std_minmax[0..1] (float32):
 * the range numpy.histogram needs for the Otsu threshold of that plane (cellulus/detect.py:88-91 ->
 * skimage.filters.threshold_otsu), so the fused predict -> detect path reads the plane once (histogram) instead
 * of twice.  init != 0 starts a new image; with init == 0 the call folds into what earlier calls (the other tiles
 * of the sample) left.  std_minmax: CLX_NOISE_MINMAX_FLOATS floats — the two results followed by scratch for the
 * per-block partials.  T <= 64. */
#define CLX_NOISE_MINMAX_FLOATS (2 + 2 * 1024)
int clx_noise_stats_minmax(const float* preds, float* out, int T, int C, long long n, float* std_minmax,
                           int init, clx_stream stream);

/* Salt / pepper noise of the infer-mode forward (cellulus/models/unet.py:75-88: `noisy[rnd <= p] = 0.5`, then 1.0):
 * rnd (T, n) uniform randoms, raw (n) the clean tile, out (T, n): out[t][i] = rnd[t][i] <= p ? (t < n_half ? 0.5 : 1.0)
 * : raw[i], the comparison in float32 as torch compares a float32 tensor with a Python scalar.  One launch in place
 * of torch's compare + where. */
int clx_noise_inject(const float* rnd, const float* raw, float* out, int T, int n_half, long long n, float p,
                     clx_stream stream);
/* Per-column minimum and maximum of a (n, width) float64 array, width 2 or 3: extent[0..width) = minima,
 * extent[width..2 width) = maxima — the bounding box the uniform grid of the mean-shift fit points is laid over
 * (sklearn builds a KD-tree there; cellulus/utils/mean_shift.py:62-74).  extent: CLX_ROWS_EXTENT_DOUBLES doubles (the
 * results + room for the block partials).  One launch up to 256 rows x 256, two above. */
#define CLX_ROWS_EXTENT_DOUBLES (6 * (1 + 256))
int clx_rows_extent_f64(const double* src, long long n, int width, double* extent, clx_stream stream);
/* Zero fills of `count` device buffers (16-byte aligned, sizes multiples of 4 bytes) in one launch per eight: the
 * accumulators a training step adds into (packed weight gradients, bias gradients, the scatter target of the loss). */
int clx_zero_many(void* const* buffers, const long long* nbytes, int count, clx_stream stream);
/* dst[i][:] = src[rows[i]][:] for float64 rows of `width` values: the random subsample MeanShift is fitted on
 * (cellulus/utils/mean_shift.py:69-70, `X[np.random.rand(len(X)) < p]`; the host draws the mask and sends the row
 * numbers, so the device-side size is known without a synchronising masked select). */
int clx_gather_rows_f64(const double* src, const int* rows, long long n, int width, double* dst, clx_stream stream);

/* The noisy copies of one tile (cellulus/models/unet.py:75-88: 2 * num_infer_iterations forwards of the same image
 * with salt / pepper noise in p_salt_pepper of its pixels) differ from the clean image only around those pixels.  Behind
 * the first k x k convolution the changed output pixels are the window-dilated set, and 1 x 1 layers keep it; a row of a
 * 1 x 1 layer depends on the same row of its input alone, so those layers are computed once on the clean image and
 * again on the CHANGED rows of each copy — the same bits as the dense computation.  The four calls below move the rows.
 *
 * clx_changed_rows: clean (C, ID, IH, IW) and noisy (T, C, ID, IH, IW), planar float32.  An output pixel of a
 * (KD, KH, KW) valid convolution of copy t is CHANGED if any input value in its window differs from the clean image's.
 * The copies are taken in chunks of `chunk`; chunk c's changed rows are written to rows[c * cap ...] as
 * (t - c * chunk) * npix_out + output pixel (any order), their number to counts[c] (which may exceed cap: rows beyond
 * cap are not written — the caller then takes the dense path).  counts: ceil(T / chunk) ints.
 * workspace: clx_changed_rows_workspace(T, ID, IH, IW) bytes of scratch (a bit per input pixel and per output row of
 * every copy), 8-byte aligned.  KW < 64. */
size_t clx_changed_rows_workspace(int T, int ID, int IH, int IW);
int clx_changed_rows(const float* clean, const float* noisy, int T, int C, int ID, int IH, int IW, int KD, int KH,
                     int KW, int chunk, int* rows, int* counts, long long cap, void* workspace, clx_stream stream);
/* After clx_changed_rows (same T, extents and window; 2-D: one output plane), from the row bits it left in `workspace`:
 * the output TILES (tile x tile) of a (WH, WW) valid convolution over those rows that see a changed row in their
 * (tile + WH - 1) x (tile + WW - 1) window — what clx_conv_desc.tile_list takes for the Winograd layer behind the 1 x 1
 * layers.  Chunk c's tiles go to tiles[c * cap ...] as ((t - c * chunk) * th + ty) * tw + tx (any order), their number to
 * counts[c] (may exceed cap; tiles beyond cap are not written).  tile + WW - 1 <= 64. */
int clx_changed_tiles(const void* workspace, int T, int ID, int IH, int IW, int KD, int KH, int KW, int WH, int WW,
                      int tile, int chunk, int* tiles, int* counts, long long cap, clx_stream stream);
/* The first layer of a ONE-channel image (3 x 3 or 3 x 3 x 3, valid; what clx_conv_fwd runs for c_real == 1) for a LIST of
 * output pixels: x planar (B, ID, IH, IW), rows[r] = b * (output pixels per image) + output pixel, wpack / bias as for
 * clx_conv_fwd of that layer; out[r][0..N) — the same fused multiply-adds in the same order as the dense kernel, so the
 * same bits (the changed rows of noisy copies never pass through a dense first-layer tensor). */
int clx_grey_rows(const float* x, int B, int ID, int IH, int IW, int KD, const int* rows, long long n, const float* wpack,
                  const float* bias, int relu, int N, float* out, int ld_out, clx_stream stream);
/* dst[r][0..width) = src[rows[r]][0..width) for r < n (row strides ld_src / ld_dst floats; width, strides % 4 == 0,
 * 16-byte aligned bases). */
int clx_gather_rows(const float* src, int ld_src, const int* rows, long long n, int width, float* dst, int ld_dst,
                    clx_stream stream);
/* dst[rows[r]][0..width) = src[r][0..width) for r < n. */
int clx_scatter_rows(const float* src, int ld_src, const int* rows, long long n, int width, float* dst, int ld_dst,
                     clx_stream stream);
/* dst[k][0..nfloats) = src[0..nfloats) for k < copies (nfloats % 4 == 0, 16-byte aligned): the clean image's rows
 * under every copy, before clx_scatter_rows overwrites the changed ones. */
int clx_broadcast_rows(const float* src, long long nfloats, float* dst, int copies, clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Mean-shift clustering (cellulus/utils/mean_shift.py:6-121 ->             */
/* sklearn.cluster.MeanShift.fit/predict), float64                          */
/* ------------------------------------------------------------------------ */
/* Adds pixel coordinates to the embedding IN PLACE (the reference mutates its
 * argument, mean_shift.py:15-32), builds the foreground mask std < threshold
 * and compacts foreground pixels in raster order:
 *   emb:  (ND, [Z,] Y, X) f64, channel 0 += x, 1 += y, 2 += z
 *   X:    (nfg, ND) f64 out,  index: (nfg) int32 raster index of each fg pixel
 *   nfg_out: device int32, number of foreground pixels
 * workspace: clx_ms_prepare_workspace(npix) bytes of scratch (16-byte aligned; no contents are expected; what the call
 * leaves — the foreground flags as one bit per pixel, the counts per tile and per chunk of tiles, the points in front of
 * every tile — is what clx_ms_assign_dense reads back).  Two launches: flags + counts from the std plane; the scatter
 * pass, which sums the counts in front of each tile itself.  index may be NULL (clx_ms_assign_dense does not need it). */
size_t clx_ms_prepare_workspace(long long npix);
int clx_ms_prepare(double* emb, const double* std, double threshold, int ND,
                   int Z, int Y, int X, double* Xout, int* index, int* nfg_out,
                   void* workspace, clx_stream stream);
/* The same compaction for the fused predict -> detect hand-off of infer() (cellulus/infer.py:69-73 without the
 * round trip through the float64 `embeddings` dataset, cellulus/predict.py:104-112 -> cellulus/detect.py:83): emb
 * and std are the network's float32 output, widened to float64 in registers — the values the staged path reads
 * back — so X and index are bit-identical to clx_ms_prepare's on the widened tensors.  emb is NOT modified (the
 * caller of this form discards the coordinate-added copy, as cellulus/detect.py:155-160 does).  Same workspace. */
int clx_ms_prepare_f32(const float* emb, const float* std, double threshold, int ND,
                       int Z, int Y, int X, double* Xout, int* index, int* nfg_out,
                       void* workspace, clx_stream stream);
/* Flat-kernel mean-shift of every seed over the fit points until
 * |shift| <= 1e-3*bandwidth or max_iter (sklearn _mean_shift_single_seed).
 * fit: (nfit, ND) f64; seeds: (nseeds, ND) f64; outputs centers (nseeds, ND)
 * f64, counts (nseeds) int32 = members within bandwidth of the final query,
 * iters (nseeds) int32. */
int clx_ms_iterate(const double* fit, int nfit, const double* seeds, int nseeds,
                   int ND, double bandwidth, int max_iter, double* centers,
                   int* counts, int* iters, clx_stream stream);
/* Same as clx_ms_iterate with the fit points bucketed in a uniform grid (for large
 * nfit): fit_sorted is sorted by cell id ((z*ny + y)*nx + x, cell coordinate =
 * floor((p - origin) / cell) per dim, cell >= bandwidth), cell_start (nx*ny*nz + 1)
 * holds each cell's first row.  `origin` is a HOST pointer to ND doubles. */
int clx_ms_iterate_grid(const double* fit_sorted, int nfit, const int* cell_start,
                        const double* origin, double cell, int nx, int ny, int nz,
                        const double* seeds, int nseeds, int ND, double bandwidth,
                        int max_iter, double* centers, int* counts, int* iters,
                        clx_stream stream);
/* Bucketing for clx_ms_iterate_grid: counting sort of the fit points by uniform-grid cell
 * (cell id = x + nx * (y + ny * z), cell index = clamp(floor((p - origin) / cell))), the points of
 * one cell in their original order (what a stable sort by cell id gives).  fit_sorted: (n, ND) f64
 * out; cell_start: (nx*ny*nz + 1) int32 out; `origin` is a HOST pointer to ND doubles;
 * workspace: clx_ms_bucket_workspace(n, nx*ny*nz) bytes. */
size_t clx_ms_bucket_workspace(int n, long long ncells);
int clx_ms_bucket(const double* fit, int n, int ND, const double* origin, double cell, int nx,
                  int ny, int nz, double* fit_sorted, int* cell_start, void* workspace,
                  clx_stream stream);
/* HOST function: the offsets of the reference's pair sampler (cellulus/datasets/zarr_dataset.py:185-198,
 * sample_offsets_within_radius) drawn from numpy's LEGACY global generator — MT19937, 32-bit outputs, masked rejection,
 * i.e. what `np.random.randint(-radius, radius + 1, size = ND * number_offsets)` called ND times produces — filtered
 * (0 < |o|^2 < radius^2) and cut to number_offsets rows, redrawn like the reference when too few survive.  key[624],
 * *pos: the generator's state (np.random.get_state()[1:3]), advanced exactly as numpy would have; offsets:
 * (number_offsets, ND) int64 out; *rounds (may be NULL): how many times everything was drawn.  Same values, same
 * final state as the numpy form (tests/test_cpu_host.py), a fifth of its time. */
int clx_sample_offsets_mt19937(unsigned int* key, int* pos, int radius, int ND, long long number_offsets,
                               long long* offsets, int* rounds);
/* HOST function (host pointers, no stream): sklearn MeanShift.fit's post-processing of the converged seeds —
 * identical centre tuples collapse (the last count wins), sort by (count, centre) descending, greedy removal of every
 * centre within `bandwidth` (squared distance <= bandwidth^2) of a kept one
 * [cellulus/utils/mean_shift.py:62-74 -> sklearn/cluster/_mean_shift.py MeanShift.fit].  centers (n, ND) and counts (n)
 * are what clx_ms_iterate* produced, copied to the host; out: room for n centres; *n_out: the kept ones, in sklearn's
 * order (0 when no seed had a neighbour — sklearn raises there, and so does the Python caller). */
int clx_ms_dedup_centers(const double* centers, const int* counts, int n, int ND, double bandwidth,
                         double* out, int* n_out);
/* labels[index[i]] = 1 + argmin_k |X[i] - centers[k]|  (first minimum);
 * labels (npix) int32 must be zero-filled by the caller (background = 0). */
int clx_ms_assign(const double* X, const int* index, int nfg,
                  const double* centers, int ncenters, int ND, int* labels,
                  clx_stream stream);
/* The same result with the centres bucketed into a uniform grid (order: centre ids sorted by
 * cell, cell_start: (nx*ny*nz + 1) offsets into it, cell index = clamp(floor((c - origin) /
 * cell))): a pixel examines the 3^ND cells around it and falls back to all centres only when
 * no candidate is closer than `cell`.  `origin` is a HOST pointer to ND doubles. */
int clx_ms_assign_grid(const double* X, const int* index, int nfg, const double* centers,
                       int ncenters, int ND, const int* order, const int* cell_start,
                       const double* origin, double cell, int nx, int ny, int nz, int* labels,
                       clx_stream stream);
/* The grid search with the centres handed over IN CELL ORDER (centers_sorted[j] = centre order[j]; 16-byte aligned
 * in 2-D): a candidate is one load from a contiguous range instead of two dependent ones, the ranges of the
 * neighbouring rows are fetched together — the same labels (same arithmetic, same tie rule) in about half the time
 * (replaces the nearest-centre `predict` of sklearn MeanShift, cellulus/utils/mean_shift.py:93-104). */
int clx_ms_assign_cells(const double* X, const int* index, int nfg, const double* centers_sorted,
                        int ncenters, int ND, const int* order, const int* cell_start,
                        const double* origin, double cell, int nx, int ny, int nz, int* labels,
                        clx_stream stream);
/* The same assignment writing the WHOLE label map — every pixel once, in full lines, background as 0 — instead of
 * scattering one label per foreground pixel into a map the caller zeroed.  It reads the compaction's flag words and
 * per-tile point counts back from `prepare_workspace`: the workspace the clx_ms_prepare (prepared_from_f32 = 0) or
 * clx_ms_prepare_f32 (= 1) call that produced X for this (Z, Y, X) image left behind, untouched since (the ONE piece of
 * state between two calls of this library; the raster index is not read and may have been left out of that call).
 * labels: (Z, Y, X) int32, no initial contents expected.  Same labels as clx_ms_assign_cells + a zero fill
 * (cellulus/utils/mean_shift.py:93-104 with the `-1 -> 0` of :29-32). */
int clx_ms_assign_dense(const double* X, const double* centers_sorted, int ncenters, int ND, const int* order,
                        const int* cell_start, const double* origin, double cell, int nx, int ny, int nz,
                        const void* prepare_workspace, int prepared_from_f32, int Z, int Y, int Xdim, int* labels,
                        clx_stream stream);

/* Seeds for use_seeds = true (cellulus/detect.py:128-132), float64, the libraries' operation order:
 *   clx_offset_magnitude    np.linalg.norm(emb[:ND], axis=0): emb (ND, npix) -> out (npix)
 *   clx_gaussian_filter_f64 scipy.ndimage.gaussian_filter(in, sigma) with mode "reflect": `weights`
 *                           (DEVICE, radius + 1 doubles: centre first) are the normalised kernel scipy
 *                           builds (the caller computes them as scipy does); axes in scipy's order;
 *                           in / out / tmp: three distinct npix-buffers
 *   clx_negate_f64          out = -in
 *   clx_peak_local_max      skimage.feature.peak_local_max(img) defaults: pixels equal to the maximum
 *                           of their 3^ND neighbourhood, > min(img) (minmax[0], DEVICE, e.g. from
 *                           clx_minmax_f64), not on the image border; raster indices appended to
 *                           `peaks` in arrival order, *npeaks = how many (may exceed capacity:
 *                           only the first `capacity` are stored) */
int clx_offset_magnitude(const double* emb, double* out, int ND, long long npix, clx_stream stream);
int clx_gaussian_filter_f64(const double* in, double* out, double* tmp, int Z, int Y, int X,
                            const double* weights, int radius, clx_stream stream);
int clx_negate_f64(const double* in, double* out, long long n, clx_stream stream);
int clx_peak_local_max(const double* img, int Z, int Y, int X, const double* minmax, int* peaks,
                       int capacity, int* npeaks, clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Greedy clustering (cellulus/utils/greedy_cluster.py:46-120,176-253)      */
/* ------------------------------------------------------------------------ */
/* emb: (ND, n) embeddings + coordinates of the n foreground pixels (raster order);
 * seedmap: (n) normalised seediness; both float32 (is_f64 = 0, the reference's 2-D
 * class) or float64 (is_f64 = 1, its 3-D class).  Runs the reference's whole
 * seed loop on the device: pick the unclustered pixel of highest seediness (stop
 * below seed_thresh), propose exp(-|e - c|^2 / (2 bw^2)) > 0.5, accept the
 * proposal as instance `count` if it has more than min_object_size pixels of
 * which more than half were unclustered, clear it, repeat while more than
 * min_unclustered_sum pixels remain.  instance: (n) int32 ids (0 = none);
 * result[0] = instances, result[1] = seeds tried.  workspace: 2*(n+16) bytes. */
int clx_greedy_cluster(const void* emb, const void* seedmap, int n, int ND, int is_f64,
                       double bandwidth, int min_object_size, double seed_thresh,
                       int min_unclustered_sum, void* workspace, int* instance, int* result,
                       clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Connected components + size filter (cellulus/utils/misc.py:11-25 ->      */
/* skimage.measure.label, full connectivity, background 0)                  */
/* ------------------------------------------------------------------------ */
size_t clx_cc_workspace(long long npix);
/* out[p] = raster-order component id (1..ncomp) of the maximal equal-value
 * 8-/26-connected region containing p, 0 for seg[p]==0. Components with fewer
 * than min_size pixels are removed first (min_size <= 0: keep all — and, as in
 * the reference, min_size == 0 means "return seg unchanged": the caller
 * handles that case). ncomp_out: device int32. */
int clx_cc_label_filter(const int* seg, int* out, int Z, int Y, int X,
                        int min_size, int* ncomp_out, void* workspace,
                        clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Exact squared Euclidean distance transform (scipy distance_transform_edt */
/* as used by cellulus/segment.py:41-51)                                    */
/* ------------------------------------------------------------------------ */
size_t clx_edt_workspace(long long npix);
/* out[p] = min over q with in[q]==0 of |p-q|^2 (int32).  cap > 0 limits every
 * axis search to `cap` steps: values < cap^2 are exact, larger ones are only
 * guaranteed to be >= cap^2 (enough for "distance < cap" tests); cap <= 0 is the
 * full transform.  An image without any zero reproduces scipy's phantom zero at
 * index -1 of the first axis. */
int clx_edt_sq(const unsigned char* in, int* out, int Z, int Y, int X, int cap,
               void* workspace, clx_stream stream);
/* segment "cell" post-processing in one call (segment.py:41-51):
 *   d1 = edt(seg==0); grown = d1 < grow; d2 = edt(grown); seg[d2 < shrink] = 0 */
int clx_grow_shrink(int* seg, int Z, int Y, int X, int grow, int shrink,
                    void* workspace /* 2*clx_edt_workspace(npix) + npix bytes */,
                    clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Otsu histogram (skimage.filters.threshold_otsu, cellulus/detect.py:88-91) */
/* ------------------------------------------------------------------------ */
/* minmax[0..1] = min, max of x (f64, n elements).  minmax: CLX_MINMAX_DOUBLES doubles — the two results followed by
 * scratch for the per-block partials (no two blocks meet on an address: same-address float64 atomics are served one after
 * the other at the memory side, which made the kernel's time proportional to its grid). */
#define CLX_MINMAX_DOUBLES (2 + 2 * 512)
int clx_minmax_f64(const double* x, long long n, double* minmax, clx_stream stream);
/* numpy.histogram(x, bins=nbins, range=(edges[0], edges[nbins])) with the
 * caller-supplied edges (np.linspace) — counts (nbins) int64, zeroed by caller. */
int clx_histogram_f64(const double* x, long long n, const double* edges,
                      int nbins, long long* counts, clx_stream stream);
/* The same two primitives for a float32 image whose values are to be read as float64 (the network's std plane handed
 * over in device memory; the staged path reads exactly these values widened from the `embeddings` dataset):
 * minmax and counts are bit-identical to the float64 entry points on the widened image. */
int clx_minmax_f32(const float* x, long long n, double* minmax, clx_stream stream);
int clx_histogram_f32(const float* x, long long n, const double* edges, int nbins,
                      long long* counts, clx_stream stream);

/* ------------------------------------------------------------------------ */
/* "nucleus" post-processing (cellulus/segment.py:52-101): per-instance Otsu */
/* of the raw intensities + scipy.ndimage.binary_fill_holes in the bbox      */
/* ------------------------------------------------------------------------ */
typedef enum { CLX_RAW_F32 = 0, CLX_RAW_F64 = 1, CLX_RAW_I32 = 2 } clx_raw_type;
/* For every id in [1, nid): bbox[id] = {zmin,ymin,xmin,zmax,ymax,xmax} (max < 0:
 * id absent) and vkey[id] = {min,max} of raw over seg==id as order-preserving
 * keys (f32: u32 bits, sign-flipped, in the low word; f64: same on 64 bits;
 * i32: v ^ 0x80000000).  Replaces np.unique/np.where/min/max, segment.py:58-79. */
int clx_inst_stats(const int32_t* seg, const void* raw, int raw_type, int Z, int Y,
                   int X, int nid, int32_t* bbox, unsigned long long* vkey,
                   clx_stream stream);
/* All instance histograms of skimage.filters.threshold_otsu(raw[seg==id])
 * (segment.py:80-81) in one pass.  slot[id] = row of `counts` (or -1: skip).
 * Float raw: edges_or_min = [rows][nbins+1] np.linspace edges in the raw dtype,
 * numpy.histogram's index arithmetic is followed in that dtype.  Integer raw:
 * edges_or_min = int32[rows] minimum per row, one bin per value.
 * counts: u32 [rows][nbins], zeroed by the caller. */
int clx_inst_histogram(const int32_t* seg, const void* raw, int raw_type,
                       long long npix, const int32_t* slot, int nid,
                       const void* edges_or_min, int nbins, uint32_t* counts,
                       clx_stream stream);
/* For instance k (id ids[k], ascending): mask = (seg==id) & (raw > thr[k]);
 * binary_fill_holes inside bbox[id] (face connectivity, background connected to
 * the box border stays background; ndim 2 or 3 says whether the z faces of the
 * box are borders); out[p] = max(out[p], id) on the filled mask — `out` zeroed
 * by the caller (segment.py:82-101).  scratch: one byte per bounding-box voxel,
 * instance k at scratch_off[k]. */
int clx_inst_refine(const int32_t* seg, const void* raw, int raw_type, int ndim,
                    int Z, int Y, int X, const int32_t* ids, const int32_t* bbox,
                    const double* thr, const long long* scratch_off,
                    unsigned char* scratch, int n, int32_t* out, clx_stream stream);

/* ------------------------------------------------------------------------ */
/* evaluation (cellulus/evaluate.py:72-100): compute_pairwise_IoU's          */
/* #pred x #gt full-image mask passes as ONE joint histogram of id pairs     */
/* ------------------------------------------------------------------------ */
/* present[id] = 1 for every id in `labels` (np.unique, evaluate.py:73-76);
 * *bad = 1 if an id lies outside [0, nid).  present / bad zeroed by the caller. */
int clx_label_presence(const int32_t* labels, long long n, int nid,
                       int32_t* present, int32_t* bad, clx_stream stream);
/* joint[pred_row[pred[i]]][gt_col[gt[i]]] += 1 for all i: intersections
 * (evaluate.py:84-86) directly, object sizes and unions (:87-94) from its row /
 * column sums.  pred_row / gt_col: id -> row / column tables (every id present in
 * the maps must have an entry); joint: u64 [rows][ncol], zeroed by the caller. */
int clx_joint_histogram(const int32_t* pred, const int32_t* gt, long long n,
                        const int32_t* pred_row, const int32_t* gt_col, int ncol,
                        unsigned long long* joint, clx_stream stream);

/* ------------------------------------------------------------------------ */
/* Input decoding (host side): the Blosc/LZ4 chunks zarr writes by default    */
/* (docs/examples/2d/01-data.py:35-50, read by zarr_dataset.py:104-121)       */
/* ------------------------------------------------------------------------ */
/* LZ4 block decoder on HOST buffers; returns the bytes written (<= dst_capacity) or < 0. */
long long clx_lz4_decompress(const unsigned char* src, long long src_bytes, unsigned char* dst,
                             long long dst_capacity);
/* BloscLZ stream decoder (codec 0 of a Blosc chunk) on HOST buffers; same return convention. */
long long clx_blosclz_decompress(const unsigned char* src, long long src_bytes, unsigned char* dst,
                                 long long dst_capacity);
/* Writer side (HOST buffers): one Blosc chunk, LZ4 inside, byte shuffle on / off — what zarr-python writes with
 * its default compressor (zarr outputs of predict / detect / segment, cellulus/predict.py:103-110 etc.).
 * dst must hold clx_blosc_compress_bound(nbytes) bytes; returns the chunk size or < 0. */
long long clx_blosc_compress_bound(long long nbytes);
long long clx_blosc_compress_lz4(const unsigned char* src, long long nbytes, int typesize, int shuffle,
                                 unsigned char* dst, long long dst_capacity);
/* inverse of Blosc's byte shuffle for n bytes of elements of `typesize` bytes (HOST buffers) */
int clx_unshuffle_bytes(const unsigned char* src, unsigned char* dst, long long n, int typesize);

#ifdef __cplusplus
}
#endif
#endif /* CLX_H */
