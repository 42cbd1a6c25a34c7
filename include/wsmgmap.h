/* wsmgmap.h — C ABI of libwsmgmap.so: the MI355X (gfx950) kernels behind WS-MGMap's
 * per-step policy hot path.
 *
 * The reference (PeihaoChen/WS-MGMap) has no FFI: its hot path is Python calling
 * torch / cuDNN / torch_scatter operators.  Each entry point below replaces the library
 * operator(s) the reference invokes at the cited call site (paths relative to
 * /root/reference/vlnce_baselines/).  A host binds them with ctypes (see INTEGRATION.md;
 * ws-mgmap_amd/wsmgmap/_abi.py is that binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA storage); the library
 *     never allocates, frees or synchronises; `stream` is a hipStream_t (NULL = default);
 *   - return value: 0 on success, a positive hipError_t if a launch failed, negative
 *     WSMG_E* for rejected arguments;
 *   - activations of the map stack are NHWC float32 ("pixel-major": [B][H][W][C], C
 *     contiguous); conv weights are OHWI ([Cout][KH][KW][Cin]) for forward /
 *     backward-weight and IHWO ([Cin][KH][KW][Cout]) for backward-data;
 *   - all channel counts seen by the conv engine are multiples of 32 (27-class tensors are
 *     padded to 32 by the caller).
 */
#ifndef WSMGMAP_H
#define WSMGMAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wsmg_stream_t;

#define WSMG_EINVAL (-1)   /* bad size / alignment / unsupported combination */
#define WSMG_ENOMEM (-2)   /* workspace too small */

/* library / build identification ("gfx950", ABI version) */
int wsmg_abi_version(void);
const char* wsmg_build_info(void);

/* ============================ operator 1: RGB-D -> egocentric BEV ============================ */

/* ComputeSpatialLocs.forward + the sub-sampling / validity / linearisation part of
 * ProjectToGroundPlane.forward (common/rgb_mapping.py:153-176,188-216).
 * depth [B][Hd][Wd] sensor units (x depth_scale = metres). lin_idx [B][Hf*Wf] int32: y*E+x, or -1
 * for a source the reference routes to cell 0 with -1e16 (out of range / depth 0 / height filter). */
int wsmg_bev_index(const float* depth, int B, int Hd, int Wd, float depth_scale, int Hf, int Wf, int E,
                   float local_scale, int32_t* lin_idx, wsmg_stream_t stream);

/* adaptive_max_pool1d over channels (rgb_mapping.py:81-84) fused with torch_scatter.scatter_max +
 * the eps fix-up (rgb_mapping.py:206-232).  feat [B][Cf][Hf][Wf] NCHW, out [B][C][E][E] NCHW; cells
 * without a valid source are 0.  One E x E tile per (sample, channel group) lives in LDS. */
int wsmg_bev_scatter_max(const float* feat, const int32_t* lin_idx, int B, int Cf, int Hf, int Wf, int C,
                         int E, float* out, wsmg_stream_t stream);

/* RotateTensor.forward (rgb_mapping.py:239-250): affine_grid + bilinear grid_sample (zeros padding,
 * align_corners=False) with A=[[c,s,0],[-s,c,0]], c,s = cos/sin(sign*heading[b]).
 * in [B][C][E][E] (NCHW, the scatter's planes); out [B][E][E][C] (NHWC, as everything downstream). C <= 64. */
int wsmg_bev_rotate(const float* in, const float* heading, float sign, int B, int C, int E, float* out,
                    wsmg_stream_t stream);

/* Mapping.project_feat_to_map, global-map half (rgb_mapping.py:34-35,40-56): episode reset
 * (global *= mask), paste the rotated E x E projection into the centre of a G x G agent view,
 * translate it to the agent's cell (bilinear, as get_grid + grid_sample do) and max-fuse into the
 * persistent map.  global_map [P][G][G][C] (NHWC, in/out), ego_rot [B][E][E][C] (NHWC), gps [B][2], masks [B].
 * Only the (E+4)^2 window the pasted view can reach is read-modify-written. */
int wsmg_map_fuse(const float* ego_rot, float* global_map, const float* gps, const float* masks, int B,
                  int C, int E, int G, float resolution, wsmg_stream_t stream);

/* Round 3: scatter-max and the first rotation in one launch, and the fuse that consumes its output.  `wsmg_bev_scatter_rotate`
 * = wsmg_bev_scatter_max followed by wsmg_bev_rotate (rgb_mapping.py:81-84,206-232 then 239-250, called at
 * rgb_mapping.py:34-35) except that the rotated map stays in NCHW planes [B][C][E][E]: the plane of a channel is complete in LDS when
 * the scatter ends and is sampled there.  `wsmg_map_fuse_planes` = wsmg_map_fuse for that layout (same values bit for bit; the
 * global map stays [num_proc][G][G][C]).  C % 4 == 0, C <= 64, E*E*4 <= 160 KiB. */
int wsmg_bev_scatter_rotate(const float* feat, const int32_t* lin_idx, const float* heading, float sign, int B, int Cf, int Hf,
                            int Wf, int C, int E, float* out_planes, wsmg_stream_t stream);
/* Round 6 — the index launch also compacts the valid sources (75-80 % of a frame's sources are invalid, SURVEY section 7, and every
 * one of the C plane workgroups of a sample walked all Hf x Wf entries): clist [B][Hf*Wf] uint32 = (source << 16) | cell, the valid
 * entries of each block of 8192 sources packed at the block's front, cnt [B][ceil(Hf*Wf / 8192)] their counts; lin_idx as
 * wsmg_bev_index.  Needs Hf*Wf <= 65536 and E*E <= 65536.  wsmg_bev_scatter_rotate_compact: wsmg_bev_scatter_rotate from that list
 * (the scatter is a max: order-independent, bit-identical planes; rgb_mapping.py:153-232). */
int wsmg_bev_index_compact(const float* depth, int B, int Hd, int Wd, float depth_scale, int Hf, int Wf, int E, float local_scale,
                           int32_t* lin_idx, uint32_t* clist, int32_t* cnt, wsmg_stream_t stream);
int wsmg_bev_scatter_rotate_compact(const float* feat, const uint32_t* clist, const int32_t* cnt, const float* heading, float sign, int B,
                                    int Cf, int Hf, int Wf, int C, int E, float* out_planes, wsmg_stream_t stream);
int wsmg_map_fuse_planes(const float* ego_rot_planes, float* global_map, const float* gps, const float* masks, int B, int C,
                         int E, int G, float resolution, wsmg_stream_t stream);

/* Retrieval half (rgb_mapping.py:57-70): translate the global map back, crop the centre E x E,
 * rotate by +compass.  scratch [B][E][E][C] (intermediate crop) and out [B][E][E][C] are NHWC; the host
 * exposes `out` as a channels-last view of the reference's [B][C][E][E] tensor. */
int wsmg_map_retrieve(const float* global_map, const float* gps, const float* compass, int B, int C, int E,
                      int G, float resolution, float* scratch, float* out, wsmg_stream_t stream);
/* The same in ONE launch (round 3): every output element computes the four cropped values its rotation taps need in registers
 * instead of the crop going through `scratch`; bit-identical to wsmg_map_retrieve. */
int wsmg_map_retrieve_fused(const float* global_map, const float* gps, const float* compass, int B, int C, int E, int G,
                            float resolution, float* out, wsmg_stream_t stream);
/* The same in one launch through LDS (round 5): a workgroup stages the box of global-map pixels its 8 x 8 output tile can touch
 * (<= 14 x 14 pixels x a slice of <= 40 channels) and takes the 16 taps of every element from there, the tap geometry computed once
 * per (pixel, rotation tap); bit-identical to wsmg_map_retrieve.  Any C % 4 == 0.  WSMG_RETRIEVE_TRACE=1 (diagnostic) prints
 * the phase boundaries of sampled workgroups in cycles and synchronises. */
int wsmg_map_retrieve_tiled(const float* global_map, const float* gps, const float* compass, int B, int C, int E, int G,
                            float resolution, float* out, wsmg_stream_t stream);

/* ============================ operator 2: map conv / UNet decoder engine ============================ */
/* cuDNN conv2d forward / backward at map_encoder.py:19-29,94-112, mg_map_policy.py:78-100,127,130.
 * Implicit GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32): exact float32 products, k-ordered
 * accumulation.  x [B][H][W][Cin], y [B][OH][OW][Cout]; bias may be NULL. */
int wsmg_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, int B, int H, int W,
                    int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                    wsmg_stream_t stream);

/* dX of the same convolution (also the FORWARD of ConvTranspose2d, mg_map_policy.py:79).
 * dy [B][OH][OW][Cout], w_ihwo [Cin][KH][KW][Cout], dx [B][H][W][Cin]. */
int wsmg_conv2d_bwd_data(const float* dy, const float* w_ihwo, float* dx, int B, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                         wsmg_stream_t stream);

/* dW (OHWI, float32) accumulated with atomics into dw, which must be zeroed by the caller. */
int wsmg_conv2d_bwd_weight(const float* x, const float* dy, float* dw_ohwi, int B, int H, int W, int Cin,
                           int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                           wsmg_stream_t stream);

/* per-channel column sums of a [rows][C] matrix (bias gradients).  out [C] is overwritten. */
int wsmg_channel_sum(const float* x, int64_t rows, int C, float* out, double* workspace,
                     int64_t workspace_bytes, wsmg_stream_t stream);
int64_t wsmg_channel_reduce_workspace_bytes(int64_t rows, int C);

/* BatchNorm2d (+ optional residual add) (+ ReLU), training or eval statistics
 * (nn.BatchNorm2d / F.relu at map_encoder.py:10-12,21-28; torchvision BasicBlock).
 *   train=1: batch statistics in float64, running stats updated in place
 *            (momentum, unbiased variance), save_mean / save_invstd [C] written;
 *   train=0: running statistics.
 * x, y, residual: [rows][C]; y may alias x. */
int wsmg_bn_act_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, int train,
                    int relu, int64_t rows, int C, float* y, float* save_mean, float* save_invstd,
                    double* workspace, int64_t workspace_bytes, wsmg_stream_t stream);

/* backward of the above (train statistics).  dy, x, y: [rows][C]; writes dx, dgamma, dbeta and, if
 * dresidual != NULL, the gradient of the residual input (= masked dy).  y may be NULL when relu != 0 and there
 * was no residual: the ReLU mask is then recomputed from x, mean, invstd, gamma, beta (one tensor less to read
 * in each of the two passes, and y need not be kept for the backward). */
int wsmg_bn_act_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, int relu, int64_t rows, int C,
                    float* dx, float* dresidual, float* dgamma, float* dbeta, double* workspace,
                    int64_t workspace_bytes, wsmg_stream_t stream);

/* y = a + b + c of three bf16 tensors in one pass (float32 sums, one rounding): the accumulated gradient of an activation with
 * three consumers (the encoded map of mg_map_policy.py:78-100: token projection, decoder stem, decoder full-resolution branch).
 * n % 8 == 0, 16-byte aligned pointers. */
int wsmg_add3_bf16(const void* a, const void* b, const void* c, void* y, int64_t n, wsmg_stream_t stream);

/* bias-less activation helpers of the decoder (NHWC):
 *   relu:        F.relu after the bias-carrying convs of mg_map_policy.py:89-100
 *   maxpool:     MaxPool2d(3,2,1)            torchvision resnet18 child 3 (map_encoder.py:80)
 *   upsample2x:  nn.Upsample(bilinear, align_corners=True) (map_encoder.py:84)
 *   avgpool2:    F.avg_pool2d(2,2)           (mg_map_policy.py:197) */
int wsmg_relu_fwd(const float* x, float* y, int64_t n, wsmg_stream_t stream);
/* dx [B][I][C] (in place) = (dx + g [B][C] / I) * (x > 0 if relu): the attention's gradient of the map tokens merged with the
 * broadcast gradient of their mean (mg_map_policy.py:217) and masked by the ReLU of map_cated_linear (:99-100) in one pass. */
int wsmg_token_grad_merge(float* dx, const float* x, const float* g, int B, int I, int C, int relu, wsmg_stream_t stream);
int wsmg_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, wsmg_stream_t stream);
int wsmg_maxpool3x3s2_fwd(const float* x, float* y, int B, int H, int W, int C, int OH, int OW,
                          wsmg_stream_t stream);
int wsmg_maxpool3x3s2_bwd(const float* dy, const float* x, float* dx, int B, int H, int W, int C, int OH,
                          int OW, wsmg_stream_t stream);
int wsmg_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, wsmg_stream_t stream);

/* layout changes at the boundary of the NHWC engine (ego map arrives NCHW: rgb_mapping.py:86;
 * pred_sem_map leaves NCHW: mg_map_policy.py:195).  c_src / c_dst allow channel padding
 * (27 <-> 32): channels >= c_src are written as 0, channels >= c_dst are dropped. */
int wsmg_nchw_to_nhwc(const float* x, float* y, int B, int C_src, int H, int W, int C_dst,
                      wsmg_stream_t stream);
int wsmg_nhwc_to_nchw(const float* x, float* y, int B, int C_src, int H, int W, int C_dst,
                      wsmg_stream_t stream);

/* ============================ operator 3: cross-modal single-query attention ============================ */
/* MGMapNet._attn (mg_map_policy.py:173-178): logits = q.k - 1e8*mask; attn = softmax(logits*scale);
 * out = attn.v.  q [B][C]; k, v [B][I][C] (token-major, C contiguous); mask [B][I] bytes or NULL;
 * out [B][C]; attn [B][I].  C must be 256. */
int wsmg_attn_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, float scale, int B,
                  int I, int C, float* out, float* attn, wsmg_stream_t stream);

/* backward: given dout [B][C] and dattn [B][I] (NULL = zeros; the contrastive monitor differentiates
 * the weights themselves, policy.py:80-84) write dq [B][C], dk [B][I][C], dv [B][I][C]. */
/* wsmg_attn_bwd*: passing dk == dv selects the keys-are-values form (k == v): ONE gradient tensor
 * dk[i] = dl[i] q + attn[i] dout is written (used when the key projection is folded into the query). */
int wsmg_attn_bwd(const float* q, const float* k, const float* v, const float* attn, const float* dout,
                  const float* dattn, float scale, int B, int I, int C, float* dq, float* dk, float* dv,
                  wsmg_stream_t stream);

/* ============================ bf16 storage variants (BASELINE configs[1]: bf16) ============================ */
/* Same operators with activations (x, y, dy, dx, residual, k, v) stored as bf16 in HBM (`void*` = bf16
 * elements), float32 arithmetic / accumulation, float32 parameters, statistics and weight gradients.
 * Convolutions take bf16 weights (the host casts the float32 master weights once per update) and run on
 * v_mfma_f32_32x32x16_bf16; backward-weight reads its pixel-major tiles with ds_read_b64_tr_b16.
 * `out_f32` is a flag word: bit 0 makes the conv write float32 instead of bf16, bit 1 applies ReLU after the bias
 * in the epilogue (conv + folded eval-mode BN + ReLU of the frozen encoders, conv + ReLU heads), bit 2 ADDS the
 * result to the existing contents of y / dx before the ReLU (a convolution over channel-concatenated inputs is run
 * part by part over the unconcatenated tensors: conv(cat[a,b], W) = conv(a, W[:, :Ca]) + conv(b, W[:, Ca:]), which
 * replaces torch.cat at map_encoder.py:104,110 and mg_map_policy.py:99 and the slice copies of its backward). */
int wsmg_conv2d_fwd_bf16(const void* x, const void* w_ohwi, const float* bias, void* y, int out_f32, int B, int H,
                         int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                         wsmg_stream_t stream);
/* Rollout-size forward layers (the frozen RGB UNet of unet_encoder.py:68-111 at one environment: a 7 x 7 map with 512
 * channels is ONE 128-pixel tile and 144 serial k-steps on 8 of the 256 CUs): the reduction of a tile is split over
 * `ksplit` workgroups that write float32 partial sums to `part` [ksplit][B*OH*OW][Cout]; a second launch adds them in
 * split order and runs the epilogue (bias, flags as above).  wsmg_conv2d_splitk_plan: *ksplit = 1 (do not split) or the
 * split count of this layer, *part_floats = the floats `part` must hold. */
int wsmg_conv2d_splitk_plan(int B, int OH, int OW, int Cin, int Cout, int KH, int KW, int* ksplit, long long* part_floats);
int wsmg_conv2d_fwd_bf16_splitk(const void* x, const void* w_ohwi, const float* bias, void* y, int flags, int ksplit,
                                float* part, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int OH, int OW, wsmg_stream_t stream);
/* nn.ConvTranspose2d forward without a gradient (the semantic classifier's first layer, mg_map_policy.py:59-61, on the rollout
 * route): backward-data of the adjoint convolution with an epilogue bias over the Cin output channels (eval-mode BatchNorm folded
 * into weight and bias) and flags bit 0 (float32 output) / bit 1 (ReLU).  x [B][OH][OW][Cout] -> y [B][H][W][Cin], w as IHWO. */
int wsmg_conv_transpose2d_infer_bf16(const void* x, const void* w_ihwo, const float* bias, void* y, int flags, int B, int H, int W,
                                     int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, wsmg_stream_t stream);
int wsmg_conv2d_bwd_data_bf16(const void* dy, const void* w_ihwo, void* dx, int out_f32, int B, int H, int W,
                              int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                              wsmg_stream_t stream);
/* The same two launches with the train-mode BatchNorm statistics of their output taken in the epilogue (every convolution of
 * the map stack feeds a BatchNorm2d: map_encoder.py:10-12,21-28,94-112, mg_map_policy.py:80-85): stats [nslab][2][C] float64,
 * ACCUMULATED INTO (sum, sum of squares of the bf16-rounded outputs per output channel; the m-tile index picks the slab).
 * wsmg_bn_act_fwd_bf16_pre consumes and clears them, so the separate statistics pass over y never runs.  bf16 output,
 * no accumulate flag. */
int wsmg_conv2d_fwd_bf16_stats(const void* x, const void* w_ohwi, const float* bias, void* y, int out_f32, double* stats,
                               int nslab, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                               int OW, wsmg_stream_t stream);
int wsmg_conv2d_bwd_data_bf16_stats(const void* dy, const void* w_ihwo, void* dx, int out_f32, double* stats, int nslab, int B,
                                    int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                    wsmg_stream_t stream);
/* ---- round 6: the concatenation in front of map_cated_linear without a copy in either direction ------------------------------
 * (mg_map_policy.py:89-100,197,207 of the reference: torch.cat of two Conv2d + ReLU outputs, then Conv2d + ReLU.)
 * wsmg_conv2d_fwd_bf16_ex = wsmg_conv2d_fwd_bf16_stats with y_ld (0 = Cout): the pixel pitch of y in elements, so that a convolution
 * writes its output straight into its channel slice of the tensor the next convolution reads (`y` points at the slice's first
 * channel); bf16 output, Cout and y_ld multiples of 8.
 * wsmg_conv2d_bwd_data_bf16_ex = wsmg_conv2d_bwd_data_bf16 with (a) relu_y [B][H][W][Cin] bf16 or NULL: dx is masked with it before it
 * is stored (dx = g where relu_y > 0, else 0 — the fused ReLUs of the layers that PRODUCED this convolution's input, whose separate
 * mask passes then do not run); (b) dx2 / split_c (dx2 may be NULL): the input gradient leaves as its two parts — channels
 * [0, split_c) to dx [pixels][split_c], the rest to dx2 [pixels][Cin - split_c] — what torch.cat's backward hands to the two producers,
 * contiguous; split_c a multiple of 8. */
int wsmg_conv2d_bwd_data_bf16_ex(const void* dy, const void* w_ihwo, void* dx, const void* relu_y, void* dx2, int split_c, int B, int H,
                                 int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, wsmg_stream_t stream);
int wsmg_conv2d_fwd_bf16_ex(const void* x, const void* w_ohwi, const float* bias, void* y, int flags, double* stats, int nslab, int y_ld,
                            int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                            wsmg_stream_t stream);
/* Column sums of up to 16 float32 matrices in one launch (round 6): out[c] = sum_r x[r][c] — the bias gradients of the recurrent
 * core's dense layers (mg_map_policy.py:118-132,147-152; torch: a reduction kernel + a memset per tensor).  descs is a HOST array.
 * Fixed summation order: bit-reproducible. */
typedef struct {
  const float* x;
  float* out;
  int rows, cols;
} WsmgColsumDesc;
int wsmg_colsum_multi(const WsmgColsumDesc* descs, int n, wsmg_stream_t stream);
/* A list of device-to-device copies (non-overlapping) in one launch: the inputs of a captured rollout step handed into its
 * static tensors (wsmgmap.graph.GraphedAct).  Any alignment, any byte count. */
typedef struct {
  void* dst;
  const void* src;
  long long bytes;
} WsmgCopyDesc;
int wsmg_copy_multi(const WsmgCopyDesc* descs, int n, wsmg_stream_t stream);
/* Rollout-size dense layers, one row per environment (the nn.Linear calls of mg_map_policy.py:150-197 at 1-16 rows):
 * y[r][o] = act(sum_k x[r][k] w[o][k] + bias[o]) in ONE launch (a GEMM library call is bias copy + GEMM + activation);
 * x [B][K] (pool = 1) or [B][K][pool] whose mean over the last axis is the layer's input (rgb_linear's AdaptiveAvgPool1d(1) +
 * Flatten); w [O][K] as nn.Linear stores it; act 0 none, 1 ReLU, 2 tanh; float32 throughout. */
int wsmg_linear_rows(const float* x, const float* w, const float* bias, float* y, int B, int K, int O, int act, int pool,
                     wsmg_stream_t stream);
/* The heads of one rollout step (policy.py:34-56, common/distributions.py:21-29,58-71) in one launch: prog = tanh(prog_pred(f)),
 * value = critic(f), mean = fc_mean(f), action = mean (noise == NULL: the mode) or noise * exp(logstd) + mean (noise [B][A]
 * standard normals: Normal.sample()'s arithmetic), logp = sum_j Normal(mean, exp(logstd)).log_prob(action).  f [B][K]. */
int wsmg_act_heads(const float* feat, int B, int K, const float* w_prog, const float* b_prog, const float* w_mean,
                   const float* b_mean, const float* logstd, int A, const float* w_crit, const float* b_crit, const float* noise,
                   float* prog, float* value, float* action, float* logp, wsmg_stream_t stream);
/* Contrastive-monitor auxiliary loss (policy.py:72-82 of the reference): kl[b] = mean_j tg_j (log tg_j - log att_j) with
 * tg = softmax(area_resize((hi - dis) / (hi - lo), S x S) / tau); dis = gt_path [B][H][W] float32, lo / hi = device scalars (its
 * batch-global min / max), att [B][S*S].  target [B][S*S] is written for the backward call, which returns d att (the only input
 * that receives a gradient): d att_j = - gkl[b] tg_j / att_j / (S*S).  One launch each instead of 13 / 8 torch launches. */
int wsmg_path_kl_fwd(const float* dis, const float* lo, const float* hi, const float* att, int B, int H, int W, int S, float tau,
                     float* target, float* kl, wsmg_stream_t stream);
int wsmg_path_kl_bwd(const float* gkl, const float* target, const float* att, int B, int n, float* datt, wsmg_stream_t stream);

/* One Adam step over a list of float32 parameter tensors (amsgrad = False, maximize = False; weight_decay is the L2 form
 * added to the gradient), arithmetic in torch.optim.Adam's order.  Replaces the optimizer step of the reference's update
 * (torch.optim.Adam built at common_trainer.py:67-69, stepped at dagger_trainer.py:540-541): 3 launches for the policy's 102
 * live tensors instead of 15.  descs is a HOST array; bias_correction{1,2} = 1 - beta{1,2}^step, computed by the caller. */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long long n;
} WsmgAdamDesc;
int wsmg_adam_step_multi(const WsmgAdamDesc* descs, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                         double bias_correction1, double bias_correction2, wsmg_stream_t stream);
/* The same step with the step count (already incremented, one float32) read from DEVICE memory and the bias corrections computed
 * in the kernel: the form a captured HIP graph can replay (torch.optim.Adam(capturable=True) is the stock counterpart). */
int wsmg_adam_step_multi_dev(const WsmgAdamDesc* descs, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                             const float* step_dev, wsmg_stream_t stream);

/* Tests / tools: tile of the LDS-window kernel that serves 3x3 stride-1 pad-1 layers with Cout % 128 == 0, Cin % 32 == 0, Cin >= 64 and
 * B*H*W >= 65536 (0 = off -> implicit-GEMM kernel, 1 = tile chosen by shape, 256 or 512 pixels per workgroup; default 1 or
 * env WSMG_CONV_WIN3).
 * Returns the previous choice.  No reference counterpart (the reference calls torch.nn.Conv2d, map_encoder.py:29-112). */
int wsmg_conv_debug_win3_tile(int mt);
int wsmg_bn_act_fwd_bf16_pre(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, int relu, int64_t rows, int C, void* y,
                             float* save_mean, float* save_invstd, double* stats, int nslab, wsmg_stream_t stream);
int wsmg_conv2d_bwd_weight_bf16(const void* x, const void* dy, float* dw_ohwi, int B, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                wsmg_stream_t stream);
int wsmg_channel_sum_bf16(const void* x, int64_t rows, int C, float* out, double* workspace,
                          int64_t workspace_bytes, wsmg_stream_t stream);
int wsmg_bn_act_fwd_bf16(const void* x, const void* residual, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int train,
                         int relu, int64_t rows, int C, void* y, float* save_mean, float* save_invstd,
                         double* workspace, int64_t workspace_bytes, wsmg_stream_t stream);
int wsmg_bn_act_bwd_bf16(const void* dy, const void* x, const void* y, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, int relu, int64_t rows, int C,
                         void* dx, void* dresidual, float* dgamma, float* dbeta, double* workspace,
                         int64_t workspace_bytes, wsmg_stream_t stream);
int wsmg_relu_fwd_bf16(const void* x, void* y, int64_t n, wsmg_stream_t stream);
int wsmg_token_grad_merge_bf16(void* dx, const void* x, const float* g, int B, int I, int C, int relu, wsmg_stream_t stream);
int wsmg_relu_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, wsmg_stream_t stream);
int wsmg_maxpool3x3s2_fwd_bf16(const void* x, void* y, int B, int H, int W, int C, int OH, int OW,
                               wsmg_stream_t stream);
int wsmg_maxpool3x3s2_bwd_bf16(const void* dy, const void* x, void* dx, int B, int H, int W, int C, int OH,
                               int OW, wsmg_stream_t stream);
/* The same pool keeping the winning tap of every output element (idx: one byte per element, 3 ky + kx, packed four channels to a
 * word, [B][OH][OW][C/4]) and the backward that reads the taps instead of recomputing up to four windows' arg-max per input element. */
int wsmg_maxpool3x3s2_fwd_idx_bf16(const void* x, void* y, uint32_t* idx, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t stream);
int wsmg_maxpool3x3s2_bwd_idx_bf16(const void* dy, const uint32_t* idx, void* dx, int B, int H, int W, int C, int OH, int OW,
                                   wsmg_stream_t stream);
int wsmg_upsample2x_fwd_bf16(const void* x, void* y, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_upsample2x_bwd_bf16(const void* dy, void* dx, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_avgpool2_fwd_bf16(const void* x, void* y, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_avgpool2_bwd_bf16(const void* dy, void* dx, int B, int H, int W, int C, wsmg_stream_t stream);
/* float32 NCHW -> bf16 NHWC (ego map in; gradient of pred_sem_map) and bf16 NHWC -> float32 NCHW */
int wsmg_nchw_to_nhwc_bf16(const float* x, void* y, int B, int C_src, int H, int W, int C_dst,
                           wsmg_stream_t stream);
int wsmg_nhwc_to_nchw_bf16(const void* x, float* y, int B, int C_src, int H, int W, int C_dst,
                           wsmg_stream_t stream);
/* attention with bf16 keys / values (q, out, attn, dq float32; dk, dv bf16) */
int wsmg_attn_fwd_bf16(const float* q, const void* k, const void* v, const uint8_t* mask, float scale, int B,
                       int I, int C, float* out, float* attn, wsmg_stream_t stream);
int wsmg_attn_bwd_bf16(const float* q, const void* k, const void* v, const float* attn, const float* dout,
                       const float* dattn, float scale, int B, int I, int C, float* dq, void* dk, void* dv,
                       wsmg_stream_t stream);

/* Shared-set form of the attention above: the B query rows attend over U << B key / value / mask sets (one per unique
 * instruction; row b uses set row_index[b], int64) — the update path repeats every instruction T times, and the
 * reference materialises the per-row copies (mg_map_policy.py:229-232).  k_sets, v_sets [U][I][C], mask_sets [U][I].
 * Backward writes dq [B][C] and dlogits [B][I] (d loss / d logit); the caller forms the per-set gradients
 * dK_u = sum_{b in u} dlogits[b]^T q[b], dV_u = sum_{b in u} attn[b]^T dout[b] with two small batched GEMMs. */
int wsmg_attn_shared_fwd(const float* q, const float* k_sets, const float* v_sets, const uint8_t* mask_sets,
                         const int64_t* row_index, float scale, int B, int I, int C, float* out, float* attn,
                         wsmg_stream_t stream);
int wsmg_attn_shared_bwd(const float* q, const float* k_sets, const float* v_sets, const float* attn, const float* dout,
                         const float* dattn, const int64_t* row_index, float scale, int B, int I, int C, float* dq,
                         float* dlogits, wsmg_stream_t stream);
int wsmg_attn_shared_fwd_bf16(const float* q, const void* k_sets, const void* v_sets, const uint8_t* mask_sets,
                              const int64_t* row_index, float scale, int B, int I, int C, float* out, float* attn,
                              wsmg_stream_t stream);
int wsmg_attn_shared_bwd_bf16(const float* q, const void* k_sets, const void* v_sets, const float* attn, const float* dout,
                              const float* dattn, const int64_t* row_index, float scale, int B, int I, int C, float* dq,
                              float* dlogits, wsmg_stream_t stream);

/* BASELINE configs[4] as SURVEY 8d defines it — B rows, each over its OWN token set — in ONE launch (round 6): the query fold
 * q_f = q W_k (w_k [256][256] = the Conv1d weight [C_out][C_in] of mg_map_policy.py:126-127; its bias cancels in the softmax) and
 * the attention of wsmg_attn_fp8_fwd, one workgroup per row, no workspace.  q [B][256] float32, x_e4m3 [B][L][256], L <= 224;
 * q_folded [B][256] (may be NULL) receives q_f for wsmg_attn_fp8_bwd.  Replaces wsmg_attn_fp8_fold + wsmg_attn_fp8_fwd
 * (mg_map_policy.py:173-178,229-232). */
int wsmg_attn_fp8_row_fwd(const float* q, const float* w_k, const uint8_t* x_e4m3, const float* x_scale, const int* lengths,
                          float scale, int B, int L, int C, float* q_folded, float* out, float* attn, wsmg_stream_t stream);
/* fp8 (OCP e4m3) text attention, BASELINE configs[4] (B=64, L=160), forward and backward (csrc/wsmg_attn_fp8.hip).  The k=1
 * Conv1d key projection of mg_map_policy.py:126-127 is folded into the single query — q.(W_k x_l + b_k) = (q W_k).x_l +
 * q.b_k, and the last term cancels in the softmax — so every token of x is read from HBM once, as bytes:
 *   wsmg_attn_fp8_fold   out [B][C] = in [B][C] @ W (transpose = 0: q_f = q W_k) or @ W^T (1: dq = d q_f W_k^T); W [C][C] is the
 *                        Conv1d weight [C_out][C_in]; float32 MFMA (v_mfma_f32_16x16x4_f32), exact float32 products;
 *   wsmg_attn_fp8_fwd    q_folded [B][C], x_e4m3 [B][L][C] with real value = byte value * *x_scale (device scalar), lengths [B]
 *                        (tokens >= length get the reference's -1e8 additive mask, mg_map_policy.py:175) or null -> out [B][C],
 *                        attn [B][L].  A row is split over wsmg_attn_fp8_splits(B, L) workgroups; workspace =
 *                        wsmg_attn_fp8_workspace_bytes(B, L) bytes of scratch, ticket = B zero-initialised words that every
 *                        launch leaves zeroed again;
 *   wsmg_attn_fp8_bwd    saved attn, dout [B][C], dattn [B][L] or null -> dq_folded [B][C] and dx [B][L][C] (gradient of the
 *                        de-quantised tokens, straight-through; may be null).  L <= 224.
 *   wsmg_quantize_e4m3_dev = wsmg_quantize_e4m3 with the scale read from device memory (no host round trip for amax). */
int wsmg_attn_fp8_fold(const float* in, const float* w, int B, int C, int transpose, float* out, wsmg_stream_t stream);
int wsmg_attn_fp8_splits(int B, int L);
int64_t wsmg_attn_fp8_workspace_bytes(int B, int L);
int wsmg_attn_fp8_fwd(const float* q_folded, const uint8_t* x_e4m3, const float* x_scale, const int* lengths, float scale, int B,
                      int L, int C, float* out, float* attn, void* workspace, unsigned* ticket, wsmg_stream_t stream);
int wsmg_attn_fp8_bwd(const float* q_folded, const uint8_t* x_e4m3, const float* x_scale, const float* attn, const float* dout,
                      const float* dattn, float scale, int B, int L, int C, float* dq_folded, float* dx, wsmg_stream_t stream);
int wsmg_quantize_e4m3_dev(const float* x, int64_t n, const float* scale, uint8_t* y, wsmg_stream_t stream);
/* y = e4m3(clamp(x * inv_scale, +-448)), round to nearest even; n a multiple of 4. */
int wsmg_quantize_e4m3(const float* x, int64_t n, float inv_scale, uint8_t* y, wsmg_stream_t stream);

/* wsmg_weight_relayout for n parameters in one launch per WSMG_RELAYOUT_MAX of them (descs is a HOST array, copied into the
 * kernel arguments; bf16 != 0: bf16 operands, else float32).  The 20 convolutions of the map stack (map_encoder.py:19-29,
 * 94-112, mg_map_policy.py:78-100) get their operands laid out at the start of a forward pass. */
#define WSMG_RELAYOUT_MAX 32
typedef struct {
  const float* w_oihw;   /* [O][I][KH][KW] float32 parameter */
  void* w_ohwi;          /* [O][KH][KW][I_pad] out */
  void* w_ihwo;          /* [I_pad][KH][KW][O] out, or NULL */
  int O, I, KH, KW, I_pad, reserved;
} WsmgRelayoutDesc;
int wsmg_weight_relayout_multi(const WsmgRelayoutDesc* descs, int n, int bf16, wsmg_stream_t stream);
/* Conv weight layouts of one layer in one launch: the reference's OIHW float32 parameter (checkpoint layout) ->
 * OHWI (operand of wsmg_conv2d_fwd / _bwd_weight) and, if w_ihwo != NULL, IHWO (operand of _bwd_data), input
 * channels zero-padded to I_pad; wsmg_weight_grad_to_oihw brings the OHWI float32 weight gradient back to OIHW. */
int wsmg_weight_relayout(const float* w_oihw, int O, int I, int KH, int KW, int I_pad, float* w_ohwi, float* w_ihwo,
                         wsmg_stream_t stream);
int wsmg_weight_relayout_bf16(const float* w_oihw, int O, int I, int KH, int KW, int I_pad, void* w_ohwi, void* w_ihwo,
                              wsmg_stream_t stream);
int wsmg_weight_grad_to_oihw(const float* dw_ohwi, int O, int I, int KH, int KW, int I_pad, float* dw_oihw,
                             wsmg_stream_t stream);

/* DETERMINISTIC weight gradients (the reference sets torch.backends.cudnn.deterministic = True, run.py:107-108; the weight
 * gradient itself is cuDNN's behind torch.nn.Conv2d / ConvTranspose2d, map_encoder.py:19-29,94-112, mg_map_policy.py:78-100).
 * The reduction over B*OH*OW pixels is split over `nsplit` workgroup ranges; instead of adding its partial tile into dW with
 * float atomics (arrival order: two runs differ in the last bits), every workgroup STORES it into its own slab of a workspace
 * ws [nsplit][Cout][KH][KW][Cin] float32 — nothing to zero, every element written exactly once — and
 * wsmg_weight_grad_reduce_oihw adds the slabs in an order that depends on nsplit only while it re-lays the result out as
 * the parameter's OIHW gradient (input channels >= I, the engine's padding, dropped).  Three calls per layer:
 *   1. _plan   -> nsplit and the workspace size in floats for this geometry (host only, no launch);
 *   2. _slabs  -> the partial sums (same kernels, same tiles and rates as wsmg_conv2d_bwd_weight[_bf16]); nsplit / ws_floats
 *                 are checked against the plan (WSMG_EINVAL);
 *   3. wsmg_weight_grad_reduce_oihw(ws, nsplit, O = Cout, I, KH, KW, I_pad = Cin, dw_oihw).
 * The workspace may be reused by the next layer on the same stream. */
int wsmg_conv2d_bwd_weight_bf16_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                     int* nsplit, long long* ws_floats);
int wsmg_conv2d_bwd_weight_bf16_slabs(const void* x, const void* dy, float* ws, int nsplit, long long ws_floats, int B, int H, int W,
                                      int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, wsmg_stream_t stream);
int wsmg_conv2d_bwd_weight_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                int* nsplit, long long* ws_floats);
int wsmg_conv2d_bwd_weight_slabs(const float* x, const float* dy, float* ws, int nsplit, long long ws_floats, int B, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, wsmg_stream_t stream);
int wsmg_weight_grad_reduce_oihw(const float* ws, int nsplit, int O, int I, int KH, int KW, int I_pad, float* dw_oihw,
                                 wsmg_stream_t stream);

/* torch.cat([a, b], dim=1) of the reference's NCHW tensors (UNet skip connections map_encoder.py:104,110, map
 * projections mg_map_policy.py:99) on NHWC storage: y[p] = a[p] ++ b[p] for `rows` pixels; a pixel's channel run is
 * bytes_a / bytes_b bytes (multiples of 16; any element type), 16-byte aligned pointers. */
/* Gradient of a channel concatenation read in place: `dy` rows are `ld_dy` elements apart (>= C, a multiple of 8, 16-byte aligned
 * base) — autograd returns the halves of torch.cat(..., dim=1)'s gradient (map_encoder.py:104,110, mg_map_policy.py:99) as views,
 * and the consumers below take them without a contiguous copy.  Otherwise as wsmg_bn_act_bwd / wsmg_relu_bwd. */
int wsmg_bn_act_bwd_ld(const float* dy, int64_t ld_dy, const float* x, const float* y, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, int relu, int64_t rows, int C, float* dx, float* dresidual,
                       float* dgamma, float* dbeta, double* workspace, int64_t workspace_bytes, wsmg_stream_t stream);
int wsmg_bn_act_bwd_ld_bf16(const void* dy, int64_t ld_dy, const void* x, const void* y, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_invstd, int relu, int64_t rows, int C, void* dx,
                            void* dresidual, float* dgamma, float* dbeta, double* workspace, int64_t workspace_bytes,
                            wsmg_stream_t stream);
/* Round 6, COMPUTE_DTYPE = "bf16+f32grad": the same call that also writes dx_lo = bf16(dx_f32 - bf16(dx_f32)) — the part of the float32
 * input gradient its bf16 rounding dropped.  The caller takes the producing convolution's weight gradient from (dx, dx_lo), two
 * launches of the bf16 weight-gradient kernel: a 16-mantissa-bit dY for the first layer of a backward chain (the reference trains in
 * float32 only: dagger_trainer.py:505-541). */
int wsmg_bn_act_bwd_ld_bf16_lo(const void* dy, int64_t ld_dy, const void* x, const void* y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, int relu, int64_t rows, int C, void* dx, void* dx_lo,
                               void* dresidual, float* dgamma, float* dbeta, double* workspace, int64_t workspace_bytes,
                               wsmg_stream_t stream);
int wsmg_relu_bwd_rows_bf16(const void* dy, int64_t ld_dy, const void* y, void* dx, int64_t rows, int C, wsmg_stream_t stream);
int wsmg_upsample2x_bwd_ld(const float* dy, int64_t ld_dy, float* dx, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_upsample2x_bwd_ld_bf16(const void* dy, int64_t ld_dy, void* dx, int B, int H, int W, int C, wsmg_stream_t stream);
int wsmg_cat_channels(const void* a, const void* b, void* y, int64_t rows, int bytes_a, int bytes_b, wsmg_stream_t stream);
/* y [B][2H][2W][Ca+Cb] = cat([bilinear 2x upsample (align_corners) of a [B][H][W][Ca], b [B][2H][2W][Cb]], channels), bf16, in one
 * pass: the decoder step `torch.cat([self.upsample(x), skip], dim=1)` of unet_encoder.py:95-109 / map_encoder.py:103-110 on the
 * rollout route (no gradient).  Ca, Cb multiples of 8. */
int wsmg_upsample2x_cat_bf16(const void* a, const void* b, void* y, int B, int H, int W, int Ca, int Cb, wsmg_stream_t stream);

/* per-pixel cross-entropy of the semantic-hallucination head straight from the NHWC logits (policy.py:61-66:
 * F.cross_entropy(pred_sem_map, target, reduction='none')): logits [rows][32] (classes <= 32 valid channels, the rest
 * padding), target int64 [rows]; loss [rows] = logsumexp - logit[target].  bwd: dlogits [rows][32] =
 * (softmax - onehot) * gloss[row], padded channels 0. */
int wsmg_ce_nhwc_fwd(const float* logits, const int64_t* target, int64_t rows, int classes, float* loss, wsmg_stream_t stream);
int wsmg_ce_nhwc_bwd(const float* logits, const int64_t* target, const float* gloss, int64_t rows, int classes,
                     float* dlogits, wsmg_stream_t stream);
int wsmg_ce_nhwc_fwd_bf16(const void* logits, const int64_t* target, int64_t rows, int classes, float* loss, wsmg_stream_t stream);
int wsmg_ce_nhwc_bwd_bf16(const void* logits, const int64_t* target, const float* gloss, int64_t rows, int classes,
                          void* dlogits, wsmg_stream_t stream);

/* ============================ trajectory-cache collate ============================ */
/* dagger_trainer.py:40-113 (collate_fn: time-major pad + stack over the N episodes of a batch) fused with the
 * trainer's float32 conversion (:614-617), on the device: src = device array of N device pointers to episode
 * tensors [length_n][elems] in their on-disk dtype (src_dtype 0 float16, 1 uint8, 2 int64, 3 float32, see
 * common_trainer.py:514-532), lengths = device int32 [N]; dst [T][N][elems] float32, element (t, n, :) = the
 * episode's step t if t < length_n (episodes longer than T are truncated, :82-83) else `pad` (1.0 for
 * observations, 0 for actions and weights, :85-91). */
int wsmg_collate_pad(const void* const* src, const int* lengths, int N, int T, int64_t elems, int src_dtype,
                     float pad, float* dst, wsmg_stream_t stream);
/* The same collate for the cached ego map when the map stack runs in bf16 (round 4): src episodes float16 [length_n][C][HW]
 * (common_trainer.py:514-532 stores rgb_ego_map as float16), dst bf16 [T][N][HW][C] — channels-last, i.e. the layout and
 * dtype the map encoder's first convolution reads, so the policy's NCHW float32 -> NHWC bf16 pass (1.3 GB read + 0.65 GB
 * written per update at B = 512) does not exist on the feeder route.  Values are bit-identical to wsmg_collate_pad followed by
 * wsmg_nchw_to_nhwc_bf16 (float16 -> float32 is exact, then one rounding to bf16).  C % 64 == 0, HW % 4 == 0, T * N <= 65535. */
int wsmg_collate_pad_nhwc_bf16(const void* const* src, const int* lengths, int N, int T, int C, int HW, float pad, void* dst,
                               wsmg_stream_t stream);
/* The same tensor from the SPARSE record form of the ego map (round 5; the recoded trajectory cache of wsmgmap/data/codec.py —
 * dagger_trainer.py:336-343 stores the dense float16 map, 55-80 % of which is zero): per episode n, bits[n] = uint64 [T_n][HW]
 * (bit c of pixel p's word: channel c is non-zero), off[n] = uint32 [T_n][HW] (non-zeros of the step in front of pixel p),
 * base[n] = int64 [T_n + 1] (non-zeros of the episode in front of step t), vals[n] = float16 non-zero values in (step, pixel,
 * channel) order; lengths [N]; dst bf16 [T][N][HW][64], pad for t >= lengths[n].  C == 64, T * N <= 65535. */
int wsmg_collate_ego_sparse_nhwc_bf16(const void* const* bits, const void* const* off, const void* const* base,
                                      const void* const* vals, const int* lengths, int N, int T, int C, int HW, float pad,
                                      void* dst, wsmg_stream_t stream);

/* ============================ persistent masked-GRU state encoders ============================ */
/* habitat-lab RNNStateEncoder (GRU, hidden 512) as used at mg_map_policy.py:118-123,147-152,220-227,242-249:
 * h_{t-1} is multiplied by masks[t] before every step (episode restarts), gate order r,z,n.
 * One launch runs all T <= 1023 steps: 32 cooperating workgroups (each owns its CU for the duration), W_hh in
 * registers.  gi = x W_ih^T + b_ih [T][N][3H] is computed by the caller; N <= 8.
 * sync_ws: wsmg_gru_workspace_bytes(T) bytes of 128-B-aligned device scratch, cleared by the call: control words
 * (word 1 != 0 afterwards means a wait timed out) + the step-indexed exchange image.  Forward: every h value
 * crosses workgroups as one 8-byte {value, launch-unique tag} word (flag-in-data: one memory round trip per
 * step); backward: one bounded-spin agent-scope release/acquire barrier per step.  Every word / 128-B line of the
 * image is written once per launch, by one workgroup.
 * save_*: [T][N][H] each, consumed by wsmg_gru_bwd. */
int64_t wsmg_gru_workspace_bytes(int T);
int wsmg_gru_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0, const float* masks,
                 int T, int N, int hidden, float* y, float* save_r, float* save_z, float* save_n,
                 float* save_ghn, void* sync_ws, wsmg_stream_t stream);
/* backward through time: dy [T][N][H] (gradient of every h_t), dhT [N][H] or NULL; writes dgi, dgh
 * [T][N][3H] (gradients of the input / hidden pre-activations) and dh0 [N][H].  The caller forms
 * dW_hh = dgh^T (mask * h_prev), db_hh = sum dgh, and back-propagates dgi through its input GEMM. */
int wsmg_gru_bwd(const float* dy, const float* dhT, const float* w_hh, const float* h0, const float* masks,
                 const float* y, const float* save_r, const float* save_z, const float* save_n,
                 const float* save_ghn, int T, int N, int hidden, float* dgi, float* dgh, float* dh0,
                 void* sync_ws, wsmg_stream_t stream);
/* wsmg_gru_fwd / wsmg_gru_bwd on a workspace the caller OWNS: zeroed once at allocation and used by nothing but these two entry
 * points of this process — the per-launch clear is skipped (tags are launch-unique within a process; one launch less in front of
 * each of the 16 chunk launches of the pipelined update, wsmgmap/recurrent.py).  After a reported timeout (wsmg_rnn_status != 0)
 * the owner zeroes the workspace again (the error word in it is sticky).  Concurrent launches need separate workspaces. */
int wsmg_gru_fwd_owned(const float* gi, const float* w_hh, const float* b_hh, const float* h0, const float* masks,
                       int T, int N, int hidden, float* y, float* save_r, float* save_z, float* save_n,
                       float* save_ghn, void* sync_ws, wsmg_stream_t stream);
int wsmg_gru_bwd_owned(const float* dy, const float* dhT, const float* w_hh, const float* h0, const float* masks,
                       const float* y, const float* save_r, const float* save_z, const float* save_n, const float* save_ghn,
                       int T, int N, int hidden, float* dgi, float* dgh, float* dh0, void* sync_ws, wsmg_stream_t stream);

/* ============================ persistent packed bidirectional LSTM ============================ */
/* nn.LSTM(50 -> 128, bidirectional) over packed instructions (instruction_encoder.py:80-92): row b is
 * active at token t iff t < lengths[b]; inactive positions emit 0.  Both directions run concurrently
 * in one launch (8 cooperating workgroups each, W_hh in registers, one bounded barrier per token).
 * gi [U][L][2][4H] = x W_ih^T + b_ih for (forward, reverse); w_hh [2][4H][H]; b_hh [2][4H]; U <= 8.
 * out [U][L][2H] (forward | reverse); save_gates [2][U][L][4][H], save_c [2][U][L][H] feed the backward.
 * state_ws: wsmg_lstm_workspace_bytes(L) bytes of 128-B-aligned device scratch (barrier words + exchange image). */
int64_t wsmg_lstm_workspace_bytes(int L);
int wsmg_lstm_fwd(const float* gi, const float* w_hh, const float* b_hh, const int32_t* lengths, int U, int L,
                  int hidden, float* out, float* save_gates, float* save_c, void* state_ws,
                  wsmg_stream_t stream);
/* backward through time: dout [U][L][2H] -> dgates [U][L][2][4H] (gradient of the gate pre-activations,
 * i.e. of gi and of W_hh h + b_hh); the caller forms dW_hh, db_hh and back-propagates through its GEMM. */
int wsmg_lstm_bwd(const float* dout, const float* w_hh, const int32_t* lengths, const float* save_gates,
                  const float* save_c, int U, int L, int hidden, float* dgates, void* state_ws,
                  wsmg_stream_t stream);

/* ============================ GroupNorm (frozen DD-PPO depth backbone, rollout path) ============================ */
/* nn.GroupNorm(G, C) [+ residual] [+ ReLU] on NHWC activations, inference only: x [B][HW][C] float32 (x_f32 = 1: the
 * convolution's float32 accumulators, stored unrounded) or bf16; y, residual [B][HW][C] bf16; gamma, beta [C]; statistics
 * per (sample, group) over HW x C/G elements, biased variance.  Used after every convolution of habitat-lab v0.1.5's
 * GroupNorm ResNet50 (third-party; instantiated at vlnce_baselines/models/encoders/resnet_encoders.py:25-32). */
int wsmg_group_norm_nhwc_bf16(const void* x, int x_f32, const void* residual, const float* gamma, const float* beta, int B,
                              int HW, int C, int G, float eps, int relu, void* y, wsmg_stream_t stream);

/* Status of the four persistent kernels above.  All their waits are bounded; when one times out (a cooperating
 * workgroup never became resident, e.g. CU oversubscription by another process) every workgroup leaves, the
 * kernel's slice of its outputs (y / dgi, dgh, dh0 / out / dgates) is filled with NaN, and a bit is set in a
 * process-wide word in host-mapped pinned memory: 1 gru_fwd, 2 gru_bwd, 4 lstm_fwd, 8 lstm_bwd.
 * wsmg_rnn_status returns that word WITHOUT synchronising the device (clear != 0: and resets the bits it returns);
 * a caller checks it at its next natural synchronisation point (the reference has no counterpart: cuDNN RNNs
 * cannot time out; mg_map_policy.py:220-227,242-249, instruction_encoder.py:80-92 are the replaced call sites).
 * wsmg_rnn_debug_spin_limit(n): bound every spin by n polls (0 = default, 2^20) — test hook to force a timeout.
 * Bit 16 (round 6): the grid barrier of wsmg_attn_fp8_mfma_fused timed out (its outputs are NaN).
 * wsmg_rnn_debug_inject(bits): OR `bits` into the word as a timed-out kernel would — test hook for the callers' error paths
 * (bench.py's in-process fallback, GradAllReducer's cross-rank agreement); returns the word after the OR. */
int wsmg_rnn_status(int clear);
int wsmg_rnn_debug_inject(unsigned bits);
/* Whole-sequence GRU launches chained by time chunk to kernels on other streams (round 5: the recurrent core of the update as
 * three concurrent kernel chains instead of 16 chunk launches; mg_map_policy.py:220-249).  As wsmg_gru_fwd_owned / _bwd_owned, plus:
 * steps_per_chunk divides T; in_count (may be NULL): one device counter per chunk that must reach in_target before the chunk's
 * inputs (gi, resp. dy) are read — filled by wsmg_rows_gemm_f32's `signal_count` or by the other recurrence's out_count;
 * out_count (may be NULL): one counter per chunk to which this launch adds wsmg_gru_chain_workgroups() arrivals when the chunk's
 * outputs (y and the saved gates, resp. dgi / dgh) are complete.  The caller zeroes the counters before the pass and enqueues a
 * waiter AFTER its producers. */
int wsmg_gru_fwd_chain(const float* gi, const float* w_hh, const float* b_hh, const float* h0, const float* masks,
                       int T, int N, int hidden, float* y, float* save_r, float* save_z, float* save_n,
                       float* save_ghn, void* sync_ws, int steps_per_chunk, const unsigned* in_count, unsigned in_target,
                       unsigned* out_count, wsmg_stream_t stream);
int wsmg_gru_bwd_chain(const float* dy, const float* dhT, const float* w_hh, const float* h0, const float* masks,
                       const float* y, const float* save_r, const float* save_z, const float* save_n,
                       const float* save_ghn, int T, int N, int hidden, float* dgi, float* dgh, float* dh0,
                       void* sync_ws, int steps_per_chunk, const unsigned* in_count, unsigned in_target, unsigned* out_count,
                       wsmg_stream_t stream);
int wsmg_gru_chain_workgroups(void);
int wsmg_rnn_debug_spin_limit(unsigned limit);

/* ---- the update path's heads, auxiliary-loss reduction and trainer loss (csrc/wsmg_heads.hip) ----------------------------
 * Replace, each as ONE launch per direction, the tail of tiny torch operators after the second recurrence:
 *   wsmg_update_heads_*: `pred = self.action_distribution(features).mean` (models/policy.py:96-97 -> common/distributions.py:42-57,
 *       Linear 512 -> A), `self.prog = torch.tanh(self.prog_pred(features))` (policy.py:59) and the progress monitor's per-row
 *       squared error `F.mse_loss(self.prog, observations['progress'], reduction='none').mean(-1)` (policy.py:86-88);
 *       x [B][K] float32 (K % 4 == 0), wm [A][K], bm [A], wp [K], bp [1], progress [B] or NULL; A <= 4.
 *       backward: dpred [B][A] / dprog [B] / dprog_rows [B] may each be NULL (no gradient); dx, dwm, dbm, dwp, dbp are overwritten;
 *       every sum runs in a fixed order (bit-reproducible).
 *   wsmg_aux_reduce_*: `_AuxLosses.reduce(mask)` (common/aux_losses.py:24-35): out2[0] = sum_k alpha[k] * sum over rows with
 *       mask[b] != 0 of rows[k][b], divided by the number of such rows (out2[1]); `rows` / `alpha` are HOST arrays of L <= 4
 *       device pointers / factors; masked-out rows are dropped, not multiplied (a non-finite loss on a padded row stays out).
 *       backward: drows [L][B] = mask ? daux * alpha[k] / nsel : 0.
 *   wsmg_dagger_loss_*: the trainer's loss (dagger_trainer.py:526-534): logits = tanh(pred) viewed [T][N][A]; per episode n the
 *       weighted mean over t of sum_j (logits - waypoint[..., :A])^2 with weights [T][N]; mean over n; + aux (NULL: none).
 *       out2 = {loss, action_loss}; den [N] is kept for the backward, which writes dpred [T*N][A] (d aux = d loss).
 *       waypoint rows are ld_waypoint floats apart (the reference slices `[:, :2]` of a wider tensor). */
int wsmg_update_heads_fwd(const float* x, const float* wm, const float* bm, const float* wp, const float* bp, const float* progress,
                          int B, int K, int A, float* pred, float* prog, float* prog_rows, wsmg_stream_t stream);
int wsmg_update_heads_bwd(const float* x, const float* wm, const float* wp, const float* prog, const float* progress,
                          const float* dpred, const float* dprog, const float* dprog_rows, int B, int K, int A, float* dx, float* dwm,
                          float* dbm, float* dwp, float* dbp, wsmg_stream_t stream);
int wsmg_aux_reduce_fwd(const float* const* rows, const float* alpha, int L, const unsigned char* mask, int B, float* out2,
                        wsmg_stream_t stream);
int wsmg_aux_reduce_bwd(const float* alpha, int L, const unsigned char* mask, const float* nsel, const float* daux, int B,
                        float* drows, wsmg_stream_t stream);
int wsmg_dagger_loss_fwd(const float* pred, const float* waypoint, int ld_waypoint, const float* weights, const float* aux, int T,
                         int N, int A, float* out2, float* den, wsmg_stream_t stream);
int wsmg_dagger_loss_bwd(const float* pred, const float* waypoint, int ld_waypoint, const float* weights, const float* den,
                         const float* dloss, int T, int N, int A, float* dpred, wsmg_stream_t stream);

/* ---- the semantic classifier's tail in one pass per direction (csrc/wsmg_cls_tail.hip) -----------------------------------
 * Reference: `map_classfier[4:7]` = BatchNorm2d(32) + ReLU + Conv2d(32, 27, 1) (mg_map_policy.py:78-86), the prediction monitor's
 * `F.cross_entropy(pred_sem_map, F.interpolate(gt_semantic_map, size=(48, 48)).long(), reduction='none').mean([1, 2])`
 * (policy.py:61-66) and the `AvgPool2d(2)` in front of map_classified_linear (mg_map_policy.py:93-96, 195).
 *   y2 [B][H][W][32] bf16: the 3 x 3 convolution's output (H even, W % 16 == 0); mean / invstd: its batch statistics
 *   (wsmg_bn_stats_finalize); w6 [classes][32], b6 [classes] float32 (classes <= 32); gt [B][Hg][Wg] float32 class ids or NULL
 *   (then ce_rows is NULL: no loss).
 *   forward  -> sem [B][H][W][32] bf16 logits (channels >= classes zero), pooled [B][H/2][W/2][32] bf16, ce_rows [B].
 *   backward <- g_rows [B] (gradient of ce_rows; NULL: none), dpooled (NULL: none)
 *            -> dbn: the gradient of the BatchNorm's output, ReLU mask applied (feed it to wsmg_bn_bwd_apply_bf16 with the
 *               dgamma / dbeta returned here), dgamma [32], dbeta [32], dw6 [classes][32], db6 [classes];
 *               workspace: wsmg_cls_tail_workspace_floats(B) floats.  Every sum is taken in a fixed order. */
long long wsmg_cls_tail_workspace_floats(int B);
int wsmg_cls_tail_fwd_bf16(const void* y2, const float* gamma, const float* beta, const float* mean, const float* invstd,
                           const float* w6, const float* b6, int classes, const float* gt, int Hg, int Wg, int B, int H, int W,
                           void* sem, void* pooled, float* ce_rows, wsmg_stream_t stream);
int wsmg_cls_tail_bwd_bf16(const void* y2, const float* gamma, const float* beta, const float* mean, const float* invstd,
                           const float* w6, const float* b6, int classes, const float* gt, int Hg, int Wg, const float* g_rows,
                           const void* dpooled, int B, int H, int W, void* dbn, float* workspace, long long workspace_floats,
                           float* dgamma, float* dbeta, float* dw6, float* db6, wsmg_stream_t stream);
/* BatchNorm2d (train mode) in halves (torch.nn.BatchNorm2d behind map_encoder.py:10-12 / mg_map_policy.py:80-84): statistics from
 * the slabs a convolution's epilogue filled (wsmg_conv2d_fwd_bf16_stats) -> save_mean, save_invstd, running statistics (slabs
 * returned zeroed); and the backward's apply pass alone, for a dy already masked by the ReLU and caller-supplied sums. */
int wsmg_bn_stats_finalize(double* stats, int nslab, int C, int64_t rows, float momentum, float eps, float* running_mean,
                           float* running_var, float* save_mean, float* save_invstd, wsmg_stream_t stream);
int wsmg_bn_bwd_apply_bf16(const void* dy, const void* x, const float* gamma, const float* mean, const float* invstd,
                           const float* dgamma, const float* dbeta, int64_t rows, int C, void* dx, wsmg_stream_t stream);

/* BASELINE configs[4] on the matrix cores: `_attn` (mg_map_policy.py:173-178, call site :229-232) of B rows over U shared
 * instruction sets, everything stored as OCP e4m3 bytes + one float scale per tensor (device scalars): S = Q K^T on
 * v_mfma_f32_32x32x16_fp8_fp8, float32 softmax with the reference's `- 1e8 * mask` (tokens >= lengths[u]), O = P V on the bf16
 * matrix pipe (P as a bf16 hi/lo pair, V converted exactly).  q_codes [B][256], k_codes / v_codes [U][L][256] (L <= 224),
 * row_ids [B]: the row indices grouped by set, set_start [U+1]: offsets of the groups in row_ids; out [B][256], attn [B][L]. */
int wsmg_attn_fp8_mfma_fwd(const uint8_t* q_codes, const float* q_scale, const uint8_t* k_codes, const float* k_scale,
                           const uint8_t* v_codes, const float* v_scale, const int* lengths, const int* row_ids,
                           const int* set_start, float scale, int B, int U, int L, int C, float* out, float* attn,
                           wsmg_stream_t stream);

/* The operands of wsmg_attn_fp8_mfma_fwd from float32 tensors in two launches (round 3): per-tensor scales (x_scale > 0: the
 * caller's; otherwise max(amax|x| * float32(1/448), 1e-30) — bit-equal to torch's `(x.abs().amax() / 448.0).clamp_min(1e-30)` — a NaN
 * input giving a NaN scale), the e4m3 codes of q [B][C], k_sets and v_sets
 * [U][L][C] (as wsmg_quantize_e4m3_dev rounds them), and the rows grouped by set: row_ids [B], set_start [U+1] from inverse [B]
 * (0 <= inverse[b] < U; the order of the rows inside a set is arbitrary — every row's result is its own).  scales [3] = q, k, v.
 * amax_ws: 3 words, ZERO on entry (left holding the maxima).  U <= 1024, C % 4 == 0. */
int wsmg_attn_fp8_prep(const float* q, const float* k_sets, const float* v_sets, const int64_t* inverse, int B, int U, int L, int C,
                       float q_scale, float k_scale, float v_scale, uint8_t* q_codes, uint8_t* k_codes, uint8_t* v_codes,
                       float* scales, int* row_ids, int* set_start, unsigned* amax_ws, wsmg_stream_t stream);

/* wsmg_attn_fp8_prep + wsmg_attn_fp8_mfma_fwd as ONE launch (round 5; BASELINE configs[4], mg_map_policy.py:173-178): from float32 q
 * [B][256], k_sets / v_sets [U][L][256] and inverse [B] — per-tensor scales (x_scale > 0: the caller's, else amax / 448 as above),
 * e4m3 quantisation on the fly, the rows of a set found by a ranked scan of `inverse`, S = Q K^T on the fp8 matrix pipe, float32
 * softmax, O = P V.  Results equal the two-step route bit for bit.  workspace: 16 words owned by ONE stream, zero before its first
 * use and never touched by the caller afterwards; arrivals_before: the sum of wsmg_attn_fp8_mfma_fused_arrivals() over the earlier
 * launches on this workspace THAT TOOK MAXIMA (mod 2^32); epoch: their count.  A launch with all three scales given uses neither.
 * scales_out [3] (may be NULL): the scales used.  Not capturable into a HIP graph (the arrival target is a launch argument).
 * WSMG_EINVAL when maxima are needed and U * ceil(B / 32) > 128 (the caller then takes the two-step route), L > 224, C != 256. */
int wsmg_attn_fp8_mfma_fused(const float* q, const float* k_sets, const float* v_sets, const int64_t* inverse, const int* lengths,
                             float q_scale, float k_scale, float v_scale, float scale, int B, int U, int L, int C,
                             unsigned* workspace, unsigned arrivals_before, int epoch, float* scales_out, float* out, float* attn,
                             wsmg_stream_t stream);
int wsmg_attn_fp8_mfma_fused_arrivals(int B, int U, int L, int C);

/* out[r] = mean of x[r][0..n) for R rows of n <= 160 float32 values (contiguous): `nn.AdaptiveAvgPool1d(1)` + `Flatten` in front of
 * rgb_linear (mg_map_policy.py:90-96) over the 7 x 7 positions of the RGB feature. */
int wsmg_mean_rows(const float* x, int64_t R, int n, float* out, wsmg_stream_t stream);

/* Distinct rows of an instruction-token matrix in one launch (the policy encodes every distinct instruction of a teacher-forcing
 * batch once instead of T x N times: instruction_encoder.py:68-93 / mg_map_policy.py:182 of the reference run the encoder over
 * every row).  tokens [B][L]: float32 (is_f32 != 0: integer-valued floats, as dagger_trainer.py:614-617 hands them over) or int64;
 * B <= 4096.  uniq [B][L] int64: rows 0 .. U-1 are the distinct rows in order of first appearance; inverse [B] int64: row b is
 * uniq[inverse[b]]; meta [2 + B] int64: meta[0] = U, meta[1] = the longest row (non-zero tokens), meta[2 + u] = length of uniq[u]. */
int wsmg_instruction_dedup(const void* tokens, int is_f32, int B, int L, long long* uniq, long long* inverse, long long* meta,
                           wsmg_stream_t stream);

/* ---- dense layers of the recurrent core's attention stage on a chunk of rows (csrc/wsmg_rows_gemm.hip, round 5) -------------
 * Replaces, one launch each, the nn.Linear calls between the two recurrences of MGMapNet.forward (mg_map_policy.py:229-245:
 * state_text_q_layer, text_map_q_layer (+ the folded text_map_k_layer), second_state_compress over cat(state, text_embedding,
 * map_embedding) + ReLU, the second GRU's input projection) and their backward products, at 32-512 rows, float32:
 *     C = epilogue( [A0 | A1 | A2] W^T )   w_is_kn == 0: W [N][K] row-major, row stride ldw (an nn.Linear weight; forward)
 *     C = epilogue( [A0 | A1 | A2] W   )   w_is_kn != 0: W [K][N] row-major, row stride ldw (the same weight in dX = dY W)
 * A: up to three column segments a_i [M][ka_i] (row stride lda_i; ka1 = ka2 = 0: one operand) — the reference's torch.cat;
 * C: up to three column segments c_i [M][nc_i] (row stride ldc_i) — the split of d(cat) into its parts;
 * epilogue, in this order: + bias[N] (may be NULL), + cin_i (same segmentation as C, may be NULL: beta = 1), ReLU (relu != 0),
 * zero where mask[M][N] <= 0 (row stride ldmask, may be NULL: threshold_backward of the ReLU).
 * K % 64 == 0 (K / 16 a multiple of 16, or of 4, times 1-4, 6 or 8), ka_i % 16 == 0, nc_i % 16 == 0, strides % 4 == 0; WSMG_EINVAL
 * otherwise.  Deterministic (fixed reduction order).
 * Chaining to the whole-sequence GRU launches (wsmg_gru_fwd_chain / _bwd_chain): wait_count (may be NULL): the product does not
 * read its operands before *wait_count >= wait_target (they are produced by a kernel still running on another stream) — the wait,
 * bounded, is a ONE-workgroup launch in front of the product on the same stream (a grid of spinning workgroups could keep that
 * producer off the CUs); its verdict travels through `gate_word` (one device word of the caller's, required with wait_count, written
 * by every call: nothing to initialise; one word per concurrent pass and chunk); a timeout sets bit `fail_bit` of the persistent
 * kernels' status word (wsmg_rnn_status) and fills the output with NaN.  signal_count (may be NULL): every workgroup adds one
 * arrival when its tile is stored. */
int wsmg_rows_gemm_f32(const float* a0, int lda0, int ka0, const float* a1, int lda1, int ka1, const float* a2, int lda2, int ka2,
                       const float* w, int ldw, int w_is_kn, const float* bias, const float* mask, int ldmask, int relu,
                       float* c0, int ldc0, int nc0, float* c1, int ldc1, int nc1, float* c2, int ldc2, int nc2,
                       const float* cin0, int ldcin0, const float* cin1, int ldcin1, const float* cin2, int ldcin2,
                       int M, const unsigned* wait_count, unsigned wait_target, unsigned* signal_count, int fail_bit,
                       unsigned* gate_word, wsmg_stream_t stream);
/* 1 if wsmg_rows_gemm_f32 has a launch form for a reduction of K = ka0 + ka1 + ka2, else 0 (callers gate on it: ADVICE r05) */
int wsmg_rows_gemm_supported(int K);
/* workgroups one wsmg_rows_gemm_f32 launch of M rows x N columns runs (= the arrivals it adds to signal_count) */
int wsmg_rows_gemm_workgroups(int M, int N);

/* Tests: hold `n_workgroups` whole compute units (1 024 threads + lds_bytes of LDS each) until *stop_flag != 0 (host-mapped or
 * device memory) or max_ms milliseconds have passed; `arrived` (device word, zero on entry) counts the workgroups that started.
 * What the collective library's ring kernels do to the persistent GRU / LSTM kernels' co-residency on a multi-GPU node, on a box with
 * one GPU (tests/test_gpu_round5.py).  No reference counterpart.  n_workgroups <= 256, max_ms <= 10 000. */
int wsmg_debug_occupy(int n_workgroups, int lds_bytes, int max_ms, const int* stop_flag, unsigned* arrived, wsmg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WSMGMAP_H */
