/* sfhip.h — C ABI of libsfhip.so: the MI355X (gfx950) kernels behind the SlowFast / CMDA hot path.
 *
 * The reference (weidafeng/Efficient-SlowFast) has NO FFI: its hot path is Python nn.Modules whose
 * arithmetic is executed by ATen ops.  Each entry point below therefore names the reference module
 * call(s) it replaces (file:line under SlowFast/slowfast/models/) instead of a native symbol.
 *
 * Conventions
 *   - all tensors fp32, device pointers owned by the caller, NDHWC ("channels-last 3D"):
 *     element (n,t,h,w,c) of a tensor with row pitch `cs` (floats) and channel offset `coff` lives at
 *     ((((n*T + t)*H + h)*W + w) * cs + coff + c).  A pitch larger than the channel count lets a
 *     producer write straight into a slice of a wider (concatenated) tensor — torch.cat is never run.
 *   - every function only enqueues work on `stream` (a hipStream_t passed as void*); no allocation,
 *     no synchronisation, safe to capture into a hipGraph.
 *   - return 0 on success, a negative SF_E* code on a bad argument (nothing is enqueued then).
 */
#ifndef SFHIP_H
#define SFHIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_EINVAL (-1)   /* inconsistent descriptor */
#define SF_EALIGN (-2)   /* pointer / pitch alignment not supported by any kernel variant */
#define SF_ELAUNCH (-3)  /* hipLaunch failed (hipGetLastError is left set) */
#define SF_ENOTTAKEN (-4) /* a specialised entry point (sf_conv_fwd_pw / _bx, sf_conv_wgrad_bx) does not serve this shape
                             or view: nothing was enqueued and the arguments are not at fault — call the general entry
                             point (sf_conv_fwd_ws / sf_conv_wgrad) instead */

#define SF_ACT_NONE 0
#define SF_ACT_RELU 1
#define SF_ACT_SIGMOID 2      /* head only */
#define SF_ACT_SOFTMAX 3      /* head only */
#define SF_ACT_HSIGMOID 4     /* relu6(x+3)/6, ghostnet_helper.py:27-31 */
#define SF_ACT_RELU6 5        /* min(max(x,0),6), mobilenetv2_helper.py:15-30 */

int sf_abi_version(void);
/* Name of the gfx target the library was built for ("gfx950"). */
const char* sf_build_arch(void);

/* ---- layout -------------------------------------------------------------------------------
 * NCTHW (the callers' layout, datasets/utils.py:73-112) -> NDHWC with the channel dim padded to
 * `cpad` (zeros) and the H/W borders zero-padded by ph/pw (so the stem conv needs no border test).
 * dst dims: [N, T, H+2*ph, Wp, cpad] with Wp >= W+2*pw (extra columns zero).            */
int sf_ncthw_to_ndhwc(const float* src, float* dst, int N, int C, int T, int H, int W,
                      int cpad, int ph, int pw, int Wp, void* stream);
/* NDHWC slice (pitch cs, offset coff, C channels) -> dense NCTHW (module-boundary outputs). */
int sf_ndhwc_to_ncthw(const float* src, int cs, int coff, float* dst, int N, int C, int T, int H,
                      int W, void* stream);

/* ---- dense convolution as implicit GEMM on fp32 MFMA ------------------------------------------
 * Replaces nn.Conv3d(+BatchNorm3d eval affine)(+residual add)(+ReLU) of
 *   stem_helper.py:157-178 (conv), resnet_helper.py:182-223 (a/b/c), :326-335 (branch1),
 *   video_model_builder.py:128-141 (conv_f2s), custom_video_model_builder.py:102-108
 *   (downsample_c_of_slow), wdf_attention_helper.py:21-29 (q/k/v), head Linear (head_helper.py:181).
 * out[m, n] = act( scale[n] * (sum_{tap,c} in[row(m,tap), c] * w[n, tap, c]) + bias[n] + res[m, n] )
 * Weights are pre-packed [Cout][kT*kH*kW][cin_pad] (cin_pad = Cin rounded up to 16, zero filled). */
typedef struct sf_conv_desc {
  int N, Ti, Hi, Wi, Cin;      /* input dims; Cin = contiguous floats consumed per tap            */
  int in_cs, in_coff;          /* input pitch / channel offset                                    */
  int To, Ho, Wo, Cout;        /* output dims                                                     */
  int out_cs, out_coff;        /* output pitch / channel offset                                   */
  int out_cmul;                /* output channel n is stored at out_coff + n*out_cmul (channel    */
                               /* shuffle folded into index math, shufflenetv2_helper.py:32-43)   */
  int kT, kH, kW, sT, sH, sW, pT, pH, pW, dT, dH, dW;
  int cin_pad;                 /* packed weight row length per tap                                */
  int act;                     /* SF_ACT_NONE | SF_ACT_RELU | SF_ACT_RELU6                        */
  int res_cs, res_coff;        /* residual pitch / offset (used when res != NULL)                 */
  int transposed;              /* 1: data-gradient of the conv described by k/s/p/d: "in" is dL/dz with   */
                               /* dims (Ti,Hi,Wi) = the forward OUTPUT dims, "out" is dL/dx with dims     */
                               /* (To,Ho,Wo) = the forward INPUT dims, weights packed [Cin][tap][Cout]    */
  int os_T, os_H, os_W;        /* scattered store (all <= 1: dense).  Output position (t,h,w) of this launch  */
  int oo_T, oo_H, oo_W;        /* is written (and its residual read) at (t*os_T+oo_T, h*os_H+oo_H,            */
  int ob_T, ob_H, ob_W;        /* w*os_W+oo_W) of a destination with dims ob_T x ob_H x ob_W: one residue     */
                               /* class of the data gradient of a strided conv (resnet_helper.py:179,         */
                               /* :326-335 stride-2 3x3 / 1x1) is then a DENSE small-kernel conv over dL/dz.   */
} sf_conv_desc;
int sf_conv_fwd(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                const float* bias, const float* res, float* out, void* stream);
/* Split-K schedule for short-M / long-K layers (res4 / res5: M <= 12544 positions, K = 2304..6144): the K steps of a
 * tile are shared by S workgroups that store raw partial tiles to ws [S][M][Cout]; a finish kernel sums them in split
 * order (no float atomics) and applies the epilogue.  sf_conv_fwd_ws_floats(d) = workspace floats (0: the single-pass
 * schedule is used); ws == NULL makes sf_conv_fwd_ws identical to sf_conv_fwd.                                   */
long sf_conv_fwd_ws_floats(const sf_conv_desc* d);
int sf_conv_fwd_ws(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                   const float* bias, const float* res, float* out, float* ws, void* stream);
/* nn.Conv3d(groups = G) with 1 < G < channels — the grouped 1x1x1 convs of ShuffleUnit (shufflenet_helper.py:48-63,
 * conv1x1x1(groups) :36-45, followed by channel_shuffle :22-34) and the grouped 1x3x3 of BottleneckTransform when
 * RESNET.NUM_GROUPS > 1 (resnet_helper.py:196-205) — as ONE launch: the block-diagonal GEMM's group is the grid's z
 * index of the LDS-tiled kernel (conv_igemm.hip).  `d` describes the WHOLE layer: Cin / Cout = all groups' channels,
 * cin_pad = the packed width of ONE group's Cin / G input channels; w_packed = [Cout][taps][cin_pad] = the G per-group
 * packs one after the other (sf_pack_conv_weight of the grouped parameter [Cout][Cin / G][kT][kH][kW]).  Group g reads
 * input channels [g Cin / G, (g + 1) Cin / G) and owns output channels, scale / bias / residual entries
 * [g Cout / G, (g + 1) Cout / G).  shuffle != 0: its channel j is STORED at channel j * G + g — channel_shuffle(., G)
 * as index math of the stores.  d->transposed = 1: the data gradient (d->Cin = the forward Cout, w_packed = the G
 * transposed packs [Cin_fwd][taps][pad(Cout_fwd / G)]; strided layers predicate the taps, os_* must be <= 1).
 * d->out_cmul must be 1.  groups == 1 && !shuffle: sf_conv_fwd.                                                   */
int sf_conv_fwd_grouped(const sf_conv_desc* d, int groups, int shuffle, const float* in, const float* w_packed,
                        const float* scale, const float* bias, const float* res, float* out, void* stream);
/* Training-mode BN statistics out of the conv's epilogue (Conv3d -> BatchNorm3d of stem_helper.py:239-254,
 * resnet_helper.py:150-215 in train mode): the per-wavefront conv kernels keep shifted sums of the outputs they store
 * and leave [count, K, sum(v - K), sum((v - K)^2)] x 4 channels per (part, channel quad) in stats_ws
 * [parts][Cout / 4][4][4] (a part = one M tile, or the 4 M tiles of a workgroup); sf_bn_train_stats_merge below turns those into what sf_bn_train_stats returns without a pass over
 * the activation.  sf_conv_stats_ws_floats(d) = floats of stats_ws (0: this shape never produces statistics).
 * sf_conv_fwd_stats sets *parts to the rows written, or to 0 when none were (epilogue with scale / res / act, or the
 * LDS-tiled kernels took the launch): the caller then runs sf_bn_train_stats on the output.                      */
long sf_conv_stats_ws_floats(const sf_conv_desc* d);
int sf_conv_fwd_stats(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                      const float* bias, const float* res, float* out, float* stats_ws, int* parts, void* stream);
/* Tuning knobs of the dense-conv launcher for microbenchmarks and A/B runs (process-wide, not used by the model code):
 * knob 0: value 0 routes every conv to the LDS-tiled kernels of conv_igemm.hip instead of the per-wavefront kernels of
 * conv_wave.hip; knob 1: force tile configuration `value` of conv_wave.hip (-1: planner); knob 2: force the rows per
 * M tile (0: planner); knob 3: value 1 selects 32-channel K steps; knob 4: the persistent
 * swapped-operand form (0 = never, 1 = the SF_CONV_WAVE_P level, 10 + L = level L: 1 every KS == 1 layer it covers,
 * 2 plain layers, 3 plain layers with <= 5 K steps); knob 6: conv_small.hip (bit 0 enable); knob 7: conv_bx.hip (0 off,
 * 1 where it wins, 2 every shape it covers); knob 8: its timing ablations; knobs 10 / 11 / 12: the weight-gradient kernels of
 * conv_wgrad_wave.hip — 10: value 0 routes every weight gradient to conv_wgrad.hip, 11: force the blocks per
 * wavefront (-1: planner), 12: workgroups to aim at (0: default); knob 30: the row-march depthwise kernels
 * (dwconv_march.hip) off / on, knob 31: its stride-(1,2,2) 1x3x3 / 1x5x5 marches alone; knobs 22 / 23: conv_rows.hip
 * (0 off, 1 the ring over t for 3x1x1 layers with <= 16 output channels, 2 every shape it covers) / the weight-gradient
 * ring over t of conv_wgrad_rows.hip (0 off, 1 on, -1 the environment's default).  Returns SF_EINVAL for an unknown knob. */
int sf_conv_tune(int knob, int value);
/* ---- long reductions on the bf16 matrix pipe with fp32-exact operands (conv_bx.hip) --------------------------------
 * gfx950's f32-input MFMA runs at the vector rate, its bf16 MFMA at 16x that.  Every fp32 value is the EXACT sum of
 * three bf16 pieces (8 significand bits each) and six bf16 MFMAs with fp32 accumulation give the fp32 product to fp32
 * rounding level (the dropped terms are below 2^-24 |a||b|): the 1x3x3 / 3x1x1 layers over >= 128 channels of
 * resnet_helper.py:182-223 and their data gradients run that way where it is faster than v_mfma_f32_*_f32
 * (sf_conv_fwd_ws routes them itself; sf_conv_tune(7, 0) switches it off, (7, 2) takes every shape the kernel covers).
 * sf_bx_split: x [rows][cs] fp32 (C channels from coff; C % 8 == 0) -> planes[3][rows + 1][C] bf16 (row `rows` = 0),
 *   sf_bx_planes_elems(rows, C) 16-bit elements, 16-byte aligned.
 * sf_conv_fwd_bx: sf_conv_fwd_ws with the operand planes handed in — in_planes = sf_bx_split of the input view
 *   ([N*Ti*Hi*Wi][Cin]), w_planes = sf_bx_split of the packed weights ([Cout][taps*Cin]); either may be NULL (made in
 *   ws).  ws: sf_conv_bx_ws_floats(d, have_in_planes, have_w_planes) floats; 0 = the shape is not served (SF_EINVAL). */
long sf_bx_planes_elems(long rows, int C);
int sf_bx_split(const float* x, int cs, int coff, long rows, int C, unsigned short* planes, void* stream);
/* sf_bx_split of n DENSE tensors (pitch = C) in one launch — the packed weights of every bf16-piece layer after an
 * optimizer step (models/optimizer.py step -> every nn.Conv3d weight of resnet_helper.py:182-223 changes).  items: n
 * records {const float* x; uint16_t* planes; int64_t rows; int32_t C; int32_t 0}; blk0[i] = first workgroup of item i
 * (a workgroup = 256 elements of 8 channels over (rows + 1) * C / 8), blk0[n] = nblocks.                           */
int sf_bx_split_batched(const void* items, const int* blk0, int n, int nblocks, void* stream);
/* Small-channel stride-1 "same" layers (Cin <= 32 or Cout <= 32; 1x1x1, 3x1x1, 1x3x3 — the Fast pathway's convs of
 * resnet_helper.py:182-223, the lateral and query / key / value projections) and, with desc.transposed, their data
 * gradients: conv_rows.hip (whole rows through LDS, swapped-operand MFMAs, BN statistics in the epilogue).  sf_conv_fwd
 * / sf_conv_fwd_stats route there by themselves; sf_conv_rows_parts = the statistics records per channel such a launch
 * leaves (4 per workgroup; 0: the shape is not served).  sf_conv_tune(22, 0 | 1 | 2) = off / by rule / every shape the
 * kernel covers.                                                                                                     */
int sf_conv_rows_parts(const sf_conv_desc* d);
/* Pointwise layers — 1x1x1 stride-1 convs with >= 64 channels on both sides, forward and (desc.transposed) data
 * gradient: nn.Conv3d branch2a / branch2c of resnet_helper.py:182-223 — on the bf16 matrix pipe with the ACTIVATIONS
 * split in registers (no activation planes; conv_bx.hip conv_pw_bx_kernel).  sf_conv_pw_ws_floats: workspace floats
 * (0: the shape is not served, or the launcher's time model leaves it to the f32 kernels); w_planes = sf_bx_split of the
 * packed weight ([Cout][Cin]) or NULL (made in ws).  stats != NULL (a buffer of sf_conv_pw_stats_floats(d) floats,
 * only without scale / res / activation): the training-mode BN statistics of the stored outputs as sf_conv_fwd_stats
 * leaves them — *parts record rows per channel for sf_bn_train_stats_merge.  sf_conv_fwd_ws routes here by itself
 * (sf_conv_fwd_ws_floats covers it); sf_conv_tune(21, 0 | 1 | 2) = off / by model / every shape it covers.            */
long sf_conv_pw_ws_floats(const sf_conv_desc* d, int have_w_planes);
long sf_conv_pw_stats_floats(const sf_conv_desc* d);
int sf_conv_fwd_pw(const sf_conv_desc* d, const float* in, const float* w_packed, const unsigned short* w_planes,
                   const float* scale, const float* bias, const float* res, float* out, float* ws, float* stats,
                   int* parts, void* stream);
long sf_conv_bx_ws_floats(const sf_conv_desc* d, int have_in_planes, int have_w_planes);
int sf_conv_fwd_bx(const sf_conv_desc* d, const float* in, const unsigned short* in_planes, const float* w_packed,
                   const unsigned short* w_planes, const float* scale, const float* bias, const float* res, float* out,
                   float* ws, void* stream);
/* Packs an nn.Conv3d weight [Cout][Cin][kT*kH*kW] (device) into wp [Cout][taps][cin_pad] and — when wtp != NULL —
 * wtp [Cin][taps][cout_pad] (the data-gradient order), zero padded, in one launch.                           */
int sf_pack_conv_weight(const float* w, int Cout, int Cin, int taps, float* wp, int cin_pad, float* wtp,
                        int cout_pad, void* stream);
/* The same for n weights in ONE launch (a training step re-packs every conv after the optimizer step).  items: device
 * array of n records { const float* w; float* wp; float* wtp; int Cout, Cin, taps, cin_pad, cout_pad, pad; long n_wp
 * (= Cout*taps*cin_pad), total (= n_wp + Cin*taps*cout_pad, or n_wp when wtp is NULL) } (64 bytes each); blk0: device
 * array of n + 1 ints, blk0[i] = first 256-thread workgroup of weight i, blk0[n] = nblocks.  Weight i takes, for the
 * forward layout, ceil(Cout / 8) * ceil(cin_pad / 32) workgroups when 2 <= taps <= 9 and ceil(n_wp / 256) otherwise,
 * plus ceil(Cin*taps / 32) * ceil(cout_pad / 32) for the transposed one (32 x 32 tiles through LDS; wtp required).  */
int sf_pack_conv_weights(const void* items, const int* blk0, int n, int nblocks, void* stream);

/* ---- depthwise convolution (groups == channels) ----------------------------------------------
 * ghostnet_helper.py:88-90,114-120,137-143; shufflenetv2_helper.py:62-64,74-75,89-91.
 * Weights packed [kT*kH*kW][C].  `Cout` <= C output channels are produced (GhostModule's
 * out[:, :oup] slice, ghostnet_helper.py:99).  kT x 3 x 3 (kT = 1 | 3) stride-1 "same" layers — the cheap operations
 * and stride-1 conv_dw of ghostnet_helper.py, the branch convs of shufflenetv2_helper.py — run as row marches
 * (dwconv_march.hip: a thread walks a column of rows with the 3 x 3 x kT window in registers), forward, data and weight
 * gradient; every other shape on the position-per-thread kernels.  Same results either way (fp32 sums in tap order). */
int sf_dwconv_fwd(const sf_conv_desc* d, const float* in, const float* w_packed, const float* scale,
                  const float* bias, const float* res, float* out, void* stream);

/* ---- pooling ------------------------------------------------------------------------------------
 * MaxPool3d (stem_helper.py:169-171; ShuffleNetV2 stem :248) / AvgPool3d stride 1 (head_helper.py:176).
 * is_avg: 0 = max (padding ignored), 1 = average, count_include_pad (no padding used by callers). */
typedef struct sf_pool_desc {
  int N, Ti, Hi, Wi, C, in_cs, in_coff;
  int To, Ho, Wo, out_cs, out_coff;
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int is_avg;
} sf_pool_desc;
int sf_pool_fwd(const sf_pool_desc* d, const float* in, float* out, void* stream);
/* Max pooling that also records the winning tap of every window — the FIRST maximum in (kt, kh, kw) scan order,
 * nn.MaxPool3d's rule (stem_helper.py:117-119 pool1, nonlocal_helper.py:75-79) — as one byte per output element, arg
 * [N*To*Ho*Wo][C] dense.  sf_maxpool_bwd_arg then gathers dL/dy into dL/dx from `arg` alone (no x, no y, no tie
 * search); overwrite != 0: dx is written (zero where an element won no window), else accumulated.  Both need
 * float4-addressable views (C, pitches, offsets multiples of 4, 16-byte aligned bases) and <= 255 taps: SF_EINVAL
 * otherwise (callers keep sf_pool_fwd / sf_maxpool_bwd for those).                                                 */
int sf_maxpool_fwd_arg(const sf_pool_desc* d, const float* in, float* out, unsigned char* arg, void* stream);
int sf_maxpool_bwd_arg(const sf_pool_desc* d, const unsigned char* arg, const float* dy, int dy_cs, int dy_coff,
                       float* dx, int dx_cs, int dx_coff, int overwrite, void* stream);

/* ---- CMDA Fast->Slow edge: MaxPool3d(alpha,1,1) -> ECA -> BN -> ReLU -> concat ------------------
 * custom_video_model_builder.py:131-135 + wdf_attention_helper.py:77-91, two launches:
 *  (1) pooled[b,c] = mean_{t',h,w} max_{r<alpha} x[b, t'*alpha + r, h, w, c]        (sf_tmax_mean)
 *  (2) out[b,t',h,w,coff+c] = act(scale[c] * (max_r x[...] * gate[b,c]) + bias[c])   (sf_gate_apply)
 *      gate = sigmoid(conv1d_k3(pooled)) when w3 != NULL (ECA), or
 *      gate = hsigmoid(pooled_gate[b,c]) when w3 == NULL (SqueezeExcite, ghostnet_helper.py:46-52;
 *      `pooled` then already holds conv_expand's output).                                          */
/* `ws` is caller-provided scratch of sf_tmax_mean_ws_floats(N, C) floats: the mean is reduced through a
 * FIXED number of partial sums per clip (no float atomics), so results are bit-reproducible.        */
long sf_tmax_mean_ws_floats(int N, int C);
int sf_tmax_mean(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha,
                 float* pooled, float* ws, void* stream);
int sf_gate_apply(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha,
                  const float* pooled, const float* w3, const float* scale, const float* bias, int act,
                  float* out, int out_cs, int out_coff, void* stream);

/* ---- CMDA Slow->Fast edge: SpatialAttention (flash-style, never materialises N x N) -------------
 * wdf_attention_helper.py:41-54 fused with custom_video_model_builder.py:143-146:
 *   S = q k^T (no 1/sqrt(d)), P = softmax_j(S), o = P v, y = gamma*o + x,
 *   z = relu(scale*y + bias)  (bn_s2f eval affine; skipped when scale == NULL),
 *   written `alpha` times along T (nn.Upsample nearest) into out (pitch out_cs, offset out_coff).
 * q,k,v,x: [B, N=T*H*W, C] views with pitches q_cs.. (q,k,v normally slices of one [B,N,3C] buffer).
 * gamma is read from device memory (it is an nn.Parameter).
 * o_save / lse_save (optional, training): O = P v [B, N, C] dense and the log2-domain log-sum-exp [B, N]
 * of every query row, which sf_attn_bwd recomputes P from.                                           */
int sf_attn_fwd(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                int alpha, float* o_save, float* lse_save, void* stream);

/* Same, with a workspace of sf_attn_fwd_ws_floats(B, N = T*H*W, C) floats (16-byte aligned): the launcher may then
 * cut the key sweep into up to 8 parts so that the last round of workgroups fills all CUs (B * N/128 query tiles on
 * 256 CUs otherwise leave up to one workgroup-time of tail), merging the parts' (O, max, denominator) in a second
 * kernel that also runs the epilogue.  Same results up to fp32 summation order; deterministic.              */
long sf_attn_fwd_ws_floats(int B, int N, int C);
/* Which arithmetic serves the attention products of head width C when a workspace is given (sf_attn_fwd_ws,
 * sf_attn_bwd_fused): 0 = v_mfma_f32_*_f32 (fp32 operands); 6 = every fp32 operand as the exact sum of three bf16
 * pieces and every fp32 product as six v_mfma_f32_32x32x16_bf16 (fp32 accumulate; the dropped terms are below
 * 2^-24 |a||b|, one fp32 rounding) — 17 <= C <= 64 (33..64 as two 32-channel blocks; SF_ATTN_BX64=0 keeps those on
 * the f32 MFMA) and C = 8 (packed planes) with 16-byte aligned rows, unless SF_ATTN_BX=0.                        */
int sf_attn_products_per_fp32(int C);
int sf_attn_fwd_ws(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                   const float* x, int x_cs, const float* gamma, const float* scale, const float* bias,
                   int act, float* out, int out_cs, int out_coff, int B, int T, int H, int W, int C,
                   int alpha, float* o_save, float* lse_save, float* ws, void* stream);

/* SpatialAttention backward (recompute form): dz = dL/d(gamma*O + x) [B, N, C]; dvec[i] = <dz_i, O_i>
 * (sf_rowdot); writes dq, dk, dv (overwrite).  dx = dz and dgamma = sum(dvec) are the caller's.        */
int sf_attn_bwd(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs,
                const float* dz, int dz_cs, const float* lse, const float* dvec, const float* gamma,
                float* dq, int dq_cs, float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C,
                int which /* 1 = dQ kernel, 2 = dK/dV kernel, 3 = both */, void* stream);

/* Single-sweep form of sf_attn_bwd for 16 < C <= 64: dK/dV and dQ in one pass over the (key block, query tile)
 * pairs — S and dP are recomputed once (5 MFMA products per tile instead of 3 + 4).  Each 128-key workgroup writes
 * its share of dQ to a plane of the workspace ws [B][ceil(N/128)][N][32|64] (four wavefronts summed in a fixed order
 * through LDS) and a second kernel sums the planes: no float atomics, bit-reproducible.
 * sf_attn_bwd_fused_ws_floats returns the workspace size in floats, or 0 when the shape is not served.          */
long sf_attn_bwd_fused_ws_floats(int B, int N, int C);
int sf_attn_bwd_fused(const float* q, int q_cs, const float* k, int k_cs, const float* v, int v_cs, const float* dz,
                      int dz_cs, const float* lse, const float* dvec, const float* gamma, float* dq, int dq_cs,
                      float* dk, int dk_cs, float* dv, int dv_cs, int B, int N, int C, float* ws, void* stream);

/* Which kernel instantiation sf_attn_bwd_fused launches for (B, N, C) on 16-byte aligned views, as 10 * family +
 * wavefronts per workgroup (family 1 = f32 MFMA d <= 16, 2 = packed bf16 planes d = 8, 3 = bf16 pieces d <= 32,
 * 4 = bf16 pieces d <= 64 in two channel blocks, 5 / 6 = f32 MFMA d <= 32 / <= 64, 7 = two-kernel form d = 128;
 * 0 = shape not served).  The choice depends on the batch (B * ceil(N / 256) >= 256 selects the 8-wavefront form of
 * family 3, i.e. B >= 3 at N = 25 088): parity tests assert which one they exercised.                             */
int sf_attn_bwd_variant(int B, int N, int C);
/* Process-wide knobs of the attention launchers for tests / A-B runs: knob 0 = wavefronts per workgroup of the
 * bf16-piece backward (0 = by shape, 4, 8); knob 1 = parts every sweep is cut into (0 = by fill, 1..8).          */
int sf_attn_tune(int knob, int value);

/* ---- training-mode BatchNorm3d forward pieces (batchnorm_helper.py:15-34 -> nn.BatchNorm3d, training=True)
 * sf_channel_stats: per-channel mean and BIASED variance over all rows of an NDHWC slice, reduced through
 *   a fixed number of fp32 partials combined in fp64 (bit-reproducible; ws = sf_channel_stats_ws_floats(C)).
 * sf_affine_fwd:    out = act(x * scale[c] + bias[c] + res), optionally repeated `rep` times along T
 *   (nn.Upsample nearest, custom_video_model_builder.py:120-121) — the normalise(+residual)(+ReLU) pass.   */
long sf_channel_stats_ws_floats(int C);
int sf_channel_stats(const float* x, int cs, int coff, long rows, int C, float* mean, float* var, float* ws,
                     void* stream);
/* Statistics + everything parameter-sized in one call: mean/var/invstd, scale = gamma*invstd,
 * shift = beta - mean*scale, and the in-place running_mean / running_var update (momentum, unbiased var). */
int sf_bn_train_stats(const float* x, int cs, int coff, long rows, int C, const float* gamma, const float* beta,
                      float eps, float momentum, float* run_mean, float* run_var, float* mean, float* var,
                      float* invstd, float* scale, float* shift, float* ws, void* stream);
/* The same outputs from the per-part sums of sf_conv_fwd_stats (every part shifted to one reference, summed in fp64
 * in a fixed order); C must be a multiple of 4.                                                              */
int sf_bn_train_stats_merge(const float* parts_ws, int parts, int C, const float* gamma, const float* beta, float eps,
                            float momentum, float* run_mean, float* run_var, float* mean, float* var, float* invstd,
                            float* scale, float* shift, void* stream);
int sf_affine_fwd(const float* x, int cs, int coff, int N, int T, int H, int W, int C, const float* scale,
                  const float* bias, const float* res, int res_cs, int res_coff, int act, int rep, float* out,
                  int out_cs, int out_coff, int out_cmul, void* stream);

/* ---- head tail (eval): activation over classes then mean over T,H,W -----------------------------
 * head_helper.py:217-221.  logits [B, P, K] -> out [B, K]; act = SF_ACT_SOFTMAX | SIGMOID | RELU | NONE */
int sf_head_act_mean(const float* logits, int B, int P, int K, int act, float* out, void* stream);

/* ---- elementwise helpers ------------------------------------------------------------------------
 * Copy a channel slice (concat pass-through / channel shuffle of the un-convolved half,
 * shufflenetv2_helper.py:100-107): out[.., out_coff + c*out_cmul] = in[.., in_coff + c].          */
int sf_copy_channels(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff,
                     int out_cmul, long rows, int C, void* stream);
/* channel_shuffle(x, groups) of shufflenet_helper.py:22-34 (view [B, G, C/G, ...] -> transpose(1, 2) -> flatten) in
 * one launch: out[.., out_coff + j*G + g] (+)= in[.., in_coff + g*(C/G) + j]  (accumulate != 0 adds).  Its inverse —
 * the backward's gather — is the same call with groups = C / G.                                               */
int sf_channel_shuffle(const float* in, int in_cs, int in_coff, float* out, int out_cs, int out_coff, int groups,
                       long rows, int C, int accumulate, void* stream);

/* ================================ backward (training) ============================================
 * Gradients of the ops above.  Activation gradients live in NDHWC buffers shaped like their forward
 * tensors; producers ACCUMULATE into them (the caller zero-fills once), which realises autograd's fan-in
 * sums (residual branches, the two consumers of each pathway tensor in the lateral fusions).
 *
 * Data gradient of a dense conv = sf_conv_fwd with desc.transposed = 1 (see sf_conv_desc).
 *
 * Weight gradient: partial[s][co][tap][ci] over S = sf_conv_wgrad_splits(d) position splits; the caller
 * sums the S partials (fixed order).  `d` is the FORWARD descriptor; dz is dL/d(conv output).          */
int sf_conv_wgrad_splits(const sf_conv_desc* d);
/* The same on the bf16 matrix pipe (conv_bx.hip: fp32 operands as three exact bf16 pieces, six MFMAs per product,
 * fragments by transposing LDS reads) for the long-reduction layers: sf_conv_wgrad_bx_splits(d) = S of its partial
 * buffer [S][Cout][taps][Cin] (0: shape not served, use sf_conv_wgrad); x_planes / dz_planes = sf_bx_split of the
 * input view ([N*Ti*Hi*Wi][Cin]) / of dz ([M][Cout]) or NULL (made in ws: sf_conv_wgrad_bx_ws_floats(d, have_x,
 * have_dz) floats).  sf_conv_wgrad_finish sums and un-packs the partials as for sf_conv_wgrad.  sf_conv_tune(9, 0 | 1
 * | 2): off / where it wins / every shape it covers.                                                              */
int sf_conv_wgrad_bx_splits(const sf_conv_desc* d);
long sf_conv_wgrad_bx_ws_floats(const sf_conv_desc* d, int have_x_planes, int have_dz_planes);
int sf_conv_wgrad_bx(const sf_conv_desc* d, const float* x, const unsigned short* x_planes, const float* dz, int dz_cs,
                     int dz_coff, const unsigned short* dz_planes, float* partial, float* ws, void* stream);
int sf_conv_wgrad(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff,
                  float* partial, void* stream);
/* Weight gradient of the grouped convs of sf_conv_fwd_grouped in ONE launch (group = grid z of conv_wgrad.hip's
 * LDS-tiled kernels): `d` = the forward descriptor of the whole layer, partial [S][Cout][taps][cin_pad] with S =
 * sf_conv_wgrad_grouped_splits(d, groups) (0: channels not divisible by groups); group g's rows are
 * [g Cout / G, (g + 1) Cout / G), so sf_conv_wgrad_finish(partial, S, Cout, taps, cin_pad, Cin / G, 0, dst, ...)
 * leaves nn.Conv3d's grouped parameter layout [Cout][Cin / G][kT][kH][kW].                                          */
int sf_conv_wgrad_grouped_splits(const sf_conv_desc* d, int groups);
int sf_conv_wgrad_grouped(const sf_conv_desc* d, int groups, const float* x, const float* dz, int dz_cs, int dz_coff,
                          float* partial, void* stream);
/* Sum the S split partials [S][Cout][packed_taps][cin_pad] in a fixed order and store / accumulate them in
 * nn.Conv3d's own layout dst[Cout][Cin][kT][kH][kW] (one pass, so the gradient can land directly in the
 * parameter's .grad).  fold_kw > 0: the stem layout, where a packed tap is (kt,kh) and a packed channel is
 * (kw, ci) = (c/4, c%4) (sf_conv_fwd over the NDHWC4 border-padded clip), Cin <= 4.                          */
int sf_conv_wgrad_finish(const float* partial, int S, int Cout, int packed_taps, int cin_pad, int Cin, int fold_kw,
                         float* dst, int accumulate, void* stream);

/* Training BatchNorm3d backward fused with ReLU mask (y > 0), residual fan-out (dres += g) and the sum over
 * `rep` nearest-upsampled copies:  g = sum_q dy * [y > 0];  dbeta = sum g;  dgamma = sum g * xhat;
 * dz = gamma * invstd * (g - dbeta/M - xhat * dgamma/M)  (dz may alias z).  ws: sf_bn_bwd_ws_floats(C).
 * relu: 0 none, 1 ReLU (y > 0), 2 ReLU6 (0 < y < 6), 3 = SF_BN_MASK_BYTES: `y` is the byte mask sf_affine_fwd_mask
 * left in the forward pass (cast to const float*; y_cs = C/4, y_coff = 0, rep = 1, float4-addressable views) — the
 * backward then reads 1/16 of the activation's bytes for the mask, twice.                                       */
#define SF_BN_MASK_BYTES 3
long sf_bn_bwd_ws_floats(int C);
/* sf_affine_fwd (below) for a ReLU / ReLU6 layer that also leaves mask[rows][C/4]: bit e of a byte says channel
 * 4i+e passes a gradient through the activation.  Flat float4 case only (C % 4 == 0, 16-byte addressable views, no
 * T repeat, no channel multiplier): SF_EINVAL otherwise.                                                        */
int sf_affine_fwd_mask(const float* x, int cs, int coff, int N, int T, int H, int W, int C, const float* scale,
                       const float* bias, const float* res, int res_cs, int res_coff, int act, float* out,
                       int out_cs, int out_coff, unsigned char* mask, void* stream);
int sf_bn_bwd_reduce(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                     const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep, int relu,
                     const float* mean, const float* invstd, float* dbeta, float* dgamma, float* ws, void* stream);
/* As sf_bn_bwd_reduce, additionally accumulating dbeta_acc[c] += dbeta[c], dgamma_acc[c] += dgamma[c] (the
 * BatchNorm3d bias / weight .grad) in the same launch.                                                        */
int sf_bn_bwd_reduce_acc(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                         const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep, int relu,
                         const float* mean, const float* invstd, float* dbeta, float* dgamma, float* ws,
                         float* dbeta_acc, float* dgamma_acc, void* stream);
int sf_bn_bwd_apply(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                    const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep, int relu,
                    const float* mean, const float* invstd, const float* gamma, const float* dbeta,
                    const float* dgamma, float* dz, int dz_cs, int dz_coff, float* dres, int dres_cs,
                    int dres_coff, void* stream);
/* As sf_bn_bwd_apply with dres != NULL, but dres is WRITTEN (dres = g) instead of accumulated: the first writer of
 * the residual branch's gradient buffer needs neither a zero fill nor a read of it.                         */
int sf_bn_bwd_apply_first(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                          const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int rep, int relu,
                          const float* mean, const float* invstd, const float* gamma, const float* dbeta,
                          const float* dgamma, float* dz, int dz_cs, int dz_coff, float* dres, int dres_cs,
                          int dres_coff, void* stream);

/* MaxPool3d backward (equality gather; dx accumulates).  `d` is the forward descriptor.                 */
int sf_maxpool_bwd(const sf_pool_desc* d, const float* x, const float* y, const float* dy, int dy_cs,
                   int dy_coff, float* dx, int dx_cs, int dx_coff, void* stream);
/* As sf_maxpool_bwd, but dx is WRITTEN (zero where a position won no window) instead of accumulated: the first
 * writer of the gradient buffer needs neither a zero fill nor a read of it.                                 */
int sf_maxpool_bwd_first(const sf_pool_desc* d, const float* x, const float* y, const float* dy, int dy_cs,
                         int dy_coff, float* dx, int dx_cs, int dx_coff, void* stream);

/* ECA backward: out[b,c] = sum_{t',hw} dz * max_r x (sf_tmax_dot), then
 * dx[frames holding the max] += dz * gate[b,c] + dpool[b,c]  (sf_eca_bwd_apply).  ws as sf_tmax_mean.   */
int sf_tmax_dot(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha, const float* dz,
                int dz_cs, int dz_coff, float* out, float* ws, void* stream);
int sf_eca_bwd_apply(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int alpha,
                     const float* dz, int dz_cs, int dz_coff, const float* gate, const float* dpool, float* dx,
                     int dx_cs, int dx_coff, void* stream);

/* ECA's gate algebra on [N, C] vectors — backward of Conv1d(1,1,k=3,pad=1,bias=False) along the channel axis +
 * sigmoid (reference wdf_attention_helper.py:68-69, 83-88; replaces the F.conv1d / conv_transpose1d calls autograd
 * would make): gate = sigmoid(w3 (*) pooled), da = dg*gate*(1-gate), dpool = dpool_scale * convT(da, w3),
 * dw3[k] += sum da * shift_k(pooled).  dg = sf_tmax_dot's output; gate / dpool feed sf_eca_bwd_apply.        */
int sf_eca_gate_bwd(const float* dg, const float* pooled, const float* w3, int N, int C, float dpool_scale,
                    float* gate, float* dpool, float* dw3, void* stream);

/* Depthwise conv backward (`d` = forward descriptor, weights [taps][C]): dx accumulates; dw [taps][C] through
 * a fixed-count partial workspace (sf_dwconv_wgrad_ws_floats).                                           */
int sf_dwconv_dgrad(const sf_conv_desc* d, const float* dz, int dz_cs, int dz_coff, const float* w_packed,
                    float* dx, int dx_cs, int dx_coff, int C, void* stream);
long sf_dwconv_wgrad_ws_floats(const sf_conv_desc* d, int C);
int sf_dwconv_wgrad(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, int C,
                    float* dw, float* ws, void* stream);
/* sf_dwconv_wgrad with the sum stored (accumulate == 0) or accumulated in the parameter's own layout
 * dw_param[C][1][kT][kH][kW] (nn.Conv3d(C, C, k, groups = C).weight): the gradient reaches .grad in the same launch. */
int sf_dwconv_wgrad_param(const sf_conv_desc* d, const float* x, const float* dz, int dz_cs, int dz_coff, int C,
                          float* dw_param, int accumulate, float* ws, void* stream);
/* out[r, c] (+)= in[r, in_coff + c*in_cmul]: backward of a channel-multiplier (shuffled) store.           */
int sf_gather_add(const float* in, int in_cs, int in_coff, int in_cmul, float* out, int out_cs, int out_coff,
                  long rows, int C, int accumulate, void* stream);

/* g[n, r, c] += v[n, c] * scale (global-mean backward);  out[r] = scale * <a[r,:], b[r,:]>;
 * out[r, c] (+)= alpha * a[r, c].                                                                       */
int sf_bcast_add(float* g, int cs, int coff, int N, long rows_per_n, int C, const float* v, float scale,
                 void* stream);
int sf_rowdot(const float* a, int a_cs, int a_coff, const float* b, int b_cs, int b_coff, long rows, int C,
              float scale, float* out, void* stream);
int sf_axpy(const float* a, int a_cs, int a_coff, float alpha, float* out, int out_cs, int out_coff, long rows,
            int C, int accumulate, void* stream);
/* ---- SubBatchNorm3d (batchnorm_helper.py:37-109): the batch is viewed as [N/S, S*C, T, H, W], i.e. sample n
 * belongs to split n % S and every split normalises with its own statistics (nn.BatchNorm3d(S*C, affine=False)).
 * The `_split` variants take nsplit = S and per-channel arrays of S*C entries indexed [split*C + c] (the layout of
 * split_bn.running_mean / running_var); with nsplit = 1 they are the plain functions above.  Requires N % S == 0
 * and N <= 1024.  sf_bn_bwd_apply_split divides by the rows of ONE split (M = N/S * T*H*W).                 */
int sf_bn_train_stats_split(const float* x, int cs, int coff, int N, long rows_per_sample, int C, int nsplit,
                            const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                            float* run_var, float* mean, float* var, float* invstd, float* scale, float* shift,
                            float* ws, void* stream);
int sf_affine_fwd_split(const float* x, int cs, int coff, int N, int T, int H, int W, int C, int nsplit,
                        const float* scale, const float* bias, const float* res, int res_cs, int res_coff, int act,
                        int rep, float* out, int out_cs, int out_coff, int out_cmul, void* stream);
int sf_bn_bwd_reduce_split(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                           const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int nsplit,
                           int rep, int relu, const float* mean, const float* invstd, float* dbeta, float* dgamma,
                           float* ws, void* stream);
int sf_bn_bwd_apply_split(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff,
                          const float* z, int z_cs, int z_coff, int N, int T, int H, int W, int C, int nsplit,
                          int rep, int relu, const float* mean, const float* invstd, const float* gamma,
                          const float* dbeta, const float* dgamma, float* dz, int dz_cs, int dz_coff, float* dres,
                          int dres_cs, int dres_coff, void* stream);
/* dx[r, c] (+)= dy[r, c] * [0 < y[r, c] (< 6)]: backward of a bare nn.ReLU / nn.ReLU6 (act = SF_ACT_RELU |
 * SF_ACT_RELU6) that has no BN in front of it, e.g. relu(cat([out, shortcut(x)])) in the ShuffleNet v1
 * Bottleneck (shufflenet_helper.py:76-77); the mask is taken from the activation's OUTPUT y.            */
int sf_act_bwd(const float* dy, int dy_cs, int dy_coff, const float* y, int y_cs, int y_coff, int act, float* dx,
               int dx_cs, int dx_coff, long rows, int C, int accumulate, void* stream);

/* ---- Nonlocal block core (nonlocal_helper.py:105-148).  The score matrix theta^T phi of ONE sample is small here
 * (N_q <= 6272 queries x N_k <= 1568 max-pooled keys, d = 256 / 512), so it is materialised by sf_conv_fwd with the
 * sample's phi rows as the "weights" ([N_k][d] is already the packed layout), normalised in place by the kernels
 * below ("softmax": softmax(scale * s) over the keys; "dot_product" needs no kernel: 1/N_k in the GEMM epilogue),
 * and multiplied by g with a second sf_conv_fwd (weights = g^T).  x / dp: rows of C = N_k scores, pitch cs.       */
int sf_row_softmax_fwd(float* x, int cs, int coff, long rows, int C, float scale, void* stream);
/* dp <- scale * p * (dp - <p, dp>)  (gradient w.r.t. the un-scaled scores, in place over dL/dp)                 */
int sf_row_softmax_bwd(const float* p, int p_cs, int p_coff, float* dp, int dp_cs, int dp_coff, long rows, int C,
                       float scale, void* stream);

/* ---- input step (datasets/kinetics.py:230-248 -> datasets/utils.py:298-315 tensor_normalize, :151-203
 * spatial_sampling, :73-112 pack_pathway_output; transform.py:283-337 / 359-393 / 395-423 / 425-468).
 * clip: ONE decoded clip, uint8 [T,H,W,3] on the device.  The short side is scaled bilinearly to (new_h, new_w)
 * (== (H, W): no resampling), a crop x crop window at (y0, x0) of the scaled frame is taken, optionally mirrored,
 * normalised as (u/255 - mean)/std, channel-reversed if asked, and the frames frame_idx[0..n_frames) (device ints;
 * NULL = all T frames) are written to dst [n_frames][crop+2ph][Wp][4]: the stems' NDHWC layout with the channel
 * padded to 4 and zero borders, so the stem convolution consumes it without a layout pass.  mean3/std3: HOST.   */
int sf_clip_prologue(const unsigned char* clip, int T, int H, int W, int new_h, int new_w, int y0, int x0, int crop,
                     int flip, int reverse, const float* mean3, const float* std3, const int* frame_idx,
                     int n_frames, float* dst, int ph, int pw, int Wp, void* stream);

#ifdef __cplusplus
}
#endif
#endif
