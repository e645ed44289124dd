/*
 * cgrna.h - C ABI of libcgrna.so, the MI355X (gfx950) kernel library behind the
 * CycleGAN / cycle-cGAN training step.
 *
 * The reference (guyshani/cycle-GAN-RNA) has no FFI: its hot path is the Python
 * API of GAN_functions.py running on aten ops.  Each entry point below is the
 * device arithmetic of one piece of that path; the comment on each names the
 * reference lines it replaces.  The Python package binds these with ctypes
 * (plain pointers and sizes only, no torch types).
 *
 * Conventions
 *   - all matrices are dense row-major float32 unless a parameter says otherwise;
 *     "ld*" is the leading dimension in elements
 *   - every pointer is a device pointer owned by the caller; the library never
 *     allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*)
 *   - return value: 0 on success, CG_ERR_* (negative) for a rejected argument,
 *     or a positive hipError_t from the launch
 */
#ifndef CGRNA_H
#define CGRNA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CG_OK 0
#define CG_ERR_BAD_ARG (-1)
#define CG_ERR_UNSUPPORTED (-2)
#define CG_ERR_WORKSPACE (-3)

/* GEMM epilogues */
#define CG_EPI_NONE 0   /* C = acc (+ bias)                                           */
#define CG_EPI_LEAKY 1  /* h = acc + bias; pm = (h>0 ? 1 : slope) * (E ? E : 1);      */
                        /* C = h * pm; AUX = pm (if AUX != NULL)                      */
#define CG_EPI_MUL 2    /* C = (acc + bias) * E                                       */

/* cg_gemm `dtype`: the arithmetic, and (flags, with CG_BF16 only) how the operands are stored */
#define CG_F32 0
#define CG_BF16 1
#define CG_B_IS_BF16 2   /* B is stored as bf16 (ldb in bf16 elements) */
#define CG_A_IS_BF16 4   /* A is stored as bf16 (lda in bf16 elements) */
/* flag, with either arithmetic: the last 16384 bytes of `workspace` (16-byte aligned, workspace_bytes a multiple of
 * 16 and at least 32768) are arrival counters of the output tiles: all zero before the first call that names them,
 * and all zero again when any such call has finished.  A product that splits K then combines its partial sums in
 * the same launch - the last workgroup to finish a tile adds that tile's slabs, in split order: the same bits as
 * without the flag - instead of by a second launch.  Calls that may overlap on the device must use different
 * workspaces, as for the slabs themselves. */
#define CG_WS_TILE_COUNTERS 8

int cg_version(void);

/* Number of loss-partial slots per reduction (see cg_reduce_*). */
#define CG_MAX_PARTS 1024

/*
 * C[M,N] (+)= op(A)[M,K] * op(B)[K,N], float32 accumulate on MFMA.
 *   transA = 0: A is stored [M][K];  1: A is stored [K][M]
 *   transB = 0: B is stored [K][N];  1: B is stored [N][K]
 *   dtype  = CG_F32 (v_mfma_f32_32x32x2_f32, exact float32 products) or
 *            CG_BF16 (operands rounded to bf16, multiplied with v_mfma_f32_16x16x32_bf16, float32
 *            accumulate).  With CG_BF16 alone A and B are float32 in memory and are rounded while
 *            being staged into LDS; CG_BF16 | CG_B_IS_BF16 / CG_A_IS_BF16: that operand is already
 *            stored as bf16 (a shadow copy such as the one cg_adam_amsgrad writes of the weights;
 *            half the bytes, no conversion) - its address must be a multiple of 8 bytes, its leading
 *            dimension a multiple of 4 and its contiguous extent even, else CG_ERR_UNSUPPORTED
 * Replaces nn.Linear forward (GAN_functions.py:59,62,89,92,97-98,117,120,124,
 * 165,169,172,174; transA=0, transB=1), its input gradient (0,0), its weight
 * gradient (1,0), and the second-order products of the gradient penalty
 * (:194-204).  `accumulate` != 0 adds into C (CG_EPI_NONE only).
 * workspace (may be NULL): scratch for split-K partial sums, used when the
 * output has few tiles and K is long; any size is accepted (a larger one allows
 * more splits), 16-byte aligned.
 */
int cg_gemm(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
            int M, int N, int K, int transA, int transB, int dtype,
            const float* bias, int epi, const float* E, int lde, float* aux, int ldaux,
            float slope, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/*
 * cg_gemm that also keeps a bf16 image of its result: every value stored to C (after the epilogue and the
 * accumulation) is stored, rounded to nearest even, to c_bf16 at the same index (same leading dimension ldc, in
 * elements; NULL = plain cg_gemm).  The next product of a chain of layers (GAN_functions.py:89-98, 117-125, 165-174
 * and their gradients) then reads this operand with CG_A_IS_BF16 / CG_B_IS_BF16: half the bytes, no conversion while
 * staging, the same bits as rounding the float32 values there.
 */
int cg_gemm_ex(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
               int M, int N, int K, int transA, int transB, int dtype,
               const float* bias, int epi, const float* E, int lde, float* aux, int ldaux,
               float slope, int accumulate, void* c_bf16, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Train-mode BatchNorm1d over the M rows of X (GAN_functions.py:60,63,89,93,
 * 117,121,166) fused with the activation and dropout that follow it
 * (ReLU :61,89,94,117,122; LeakyReLU(0.2)+Dropout(0.3) :167-168):
 *   mean, var (biased) per column; xhat = (x-mean)*invstd;
 *   h = gamma*xhat+beta; pm = (h>0 ? 1 : slope) * (mask ? mask : 1); Y = h*pm
 * slope = 0 is ReLU, slope = 1 no activation.  mean_out/invstd_out (N) are kept
 * for the backward; running_mean/running_var (may be NULL) are updated with
 * `momentum` and the unbiased variance; pm_out (may be NULL) receives pm.
 * Y may alias X.  ybf16 (may be NULL) receives Y rounded to bf16.
 * workspace: cg_bn_workspace_bytes(M, N) bytes of scratch for the per-chunk
 * column sums (the same rule holds for cg_bn_bwd and cg_bn_bwd_bwd).
 */
size_t cg_bn_workspace_bytes(int M, int N);
int cg_bn_fwd(float* Y, int ldy, const float* X, int ldx, int M, int N,
              const float* gamma, const float* beta, float eps,
              float* mean_out, float* invstd_out,
              float* running_mean, float* running_var, float momentum,
              float slope, const float* mask, int ldmask, float* pm_out, int ldpm,
              void* ybf16, int ldybf16, void* workspace, size_t workspace_bytes, void* stream);

/*
 * A Linear layer whose output feeds a BatchNorm (GAN_functions.py:89-90, 92-93, 117-118, 120-121: every hidden
 * layer of the generators) without the pass that combines split-K partial sums: cg_gemm_slabs is cg_gemm without
 * epilogue and accumulation that, when it splits K, leaves the partial sums in `workspace` as *n_slabs dense
 * M x N slabs and does not touch C or add the bias (*n_slabs = 1: C holds the finished product with its bias, as
 * after cg_gemm).  cg_bn_fwd_slabs is cg_bn_fwd whose input is those slabs: it adds bias + slabs in slab order (the
 * same additions in the same order as cg_gemm makes), writes the sum to X (the backward pass reads it) and the
 * normalised activation to Y.  Only shapes the one-launch BatchNorm handles (cg_bn_fwd_slabs_ok(M, N) != 0);
 * the two calls must follow each other on one stream with the same workspace.
 */
int cg_gemm_slabs(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
                  int M, int N, int K, int transA, int transB, int dtype, const float* bias,
                  void* workspace, size_t workspace_bytes, int* n_slabs, void* stream);
int cg_bn_fwd_slabs_ok(int M, int N);
int cg_bn_fwd_slabs(float* Y, int ldy, float* X, int ldx, const float* slabs, int n_slabs, const float* x_bias,
                    int M, int N, const float* gamma, const float* beta, float eps,
                    float* mean_out, float* invstd_out, float* running_mean, float* running_var,
                    float momentum, float slope, const float* mask, int ldmask, float* pm_out, int ldpm,
                    void* ybf16, int ldybf16, void* stream);

/*
 * cg_bn_fwd for callers that cannot allocate device memory or hand in a workspace (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The per-chunk column
 * sums live in a 1 MiB array inside the library's device image (one copy per GPU, held until the process ends);
 * the call makes no hipMalloc / hipFree and no allocating library call.  Same arguments, kernels and results as
 * cg_bn_fwd.  Shapes whose sums do not fit - cg_bn_workspace_bytes(M, N) > 1 MiB, i.e. more than 43690 columns -
 * return CG_ERR_WORKSPACE.  Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two
 * streams) race; calls on one stream are safe.  The first call on a device looks up the array's address
 * (hipGetSymbolAddress); make it outside a stream capture.
 */
int cg_bn_fwd_static(float* Y, int ldy, const float* X, int ldx, int M, int N,
                     const float* gamma, const float* beta, float eps,
                     float* mean_out, float* invstd_out,
                     float* running_mean, float* running_var, float momentum,
                     float slope, const float* mask, int ldmask, float* pm_out, int ldpm,
                     void* ybf16, int ldybf16, void* stream);

/*
 * cg_bn_fwd on `nseg` (1..8) equal row segments of X at once: segment s is rows
 * [s*seg_rows, (s+1)*seg_rows) and is normalised with its own batch statistics,
 * exactly as a separate call of the module on that batch would
 * (the critic is called on the real, the generated and the interpolated batch
 * in a row, GAN_functions.py:326-336, and three times per species in the
 * generator step, :347,:366-371).  mean_out/invstd_out hold nseg*N values,
 * segment-major.  The running statistics receive one update per segment, in
 * the order given by `order`: 4 bits per position, position k holds the
 * segment updated k-th (the order in which the reference makes the calls).
 * workspace: nseg * cg_bn_workspace_bytes(seg_rows, N) bytes.
 */
int cg_bn_fwd_seg(float* Y, int ldy, const float* X, int ldx, int seg_rows, int nseg,
                  unsigned order, int N, const float* gamma, const float* beta, float eps,
                  float* mean_out, float* invstd_out, float* running_mean, float* running_var,
                  float momentum, float slope, const float* mask, int ldmask,
                  float* pm_out, int ldpm, void* ybf16, int ldybf16,
                  void* workspace, size_t workspace_bytes, void* stream);

/*
 * The running-statistics update of a train-mode BatchNorm1d call (GAN_functions.py:60,63,89,93,117,121,166) on its
 * own, from the batch statistics an earlier cg_bn_fwd of the SAME input stored (mean, invstd; M rows): what a second
 * forward call on that input would do to running_mean / running_var.  For a caller that applies a network twice to
 * one batch with unchanged weights (the generator step's first pass after the critic step's, :326-327 and :344-345)
 * and reuses the deterministic part of the first call.  The batch variance is recovered as 1/invstd^2 - eps.
 */
int cg_bn_running_update(float* running_mean, float* running_var, const float* mean, const float* invstd,
                         int M, int N, float eps, float momentum, void* stream);

/*
 * Eval-mode BatchNorm1d + activation (g_m2h.eval() translation pass,
 * GAN_functions.py:461-468): Y = act(gamma*(x-rm)/sqrt(rv+eps)+beta).
 * ybf16 (may be NULL): bf16 image of Y, as in cg_bn_fwd.
 */
int cg_bn_eval(float* Y, int ldy, const float* X, int ldx, int M, int N,
               const float* gamma, const float* beta, const float* running_mean,
               const float* running_var, float eps, float slope, void* ybf16, int ldybf16, void* stream);

/*
 * Backward of cg_bn_fwd (autograd of the same lines).  g = dY * pm where pm is
 * PM if given, else (Y>0 ? 1 : slope) from the saved output Y, else 1 (both NULL).
 *   dbeta (+)= sum_m g;  dgamma (+)= sum_m g*xhat;
 *   dX = invstd*gamma*(g - mean_m(g) - xhat*mean_m(g*xhat))
 * dgamma/dbeta may be NULL (input gradient only); `accumulate` adds into them.
 * gbar_out / gxbar_out (N each, may be NULL) receive mean_m(g), mean_m(g*xhat):
 * the gradient penalty's second-order pass needs them.  dX may alias dY.
 * dxbf16 (may be NULL): bf16 image of dX (leading dimension lddxbf16), for the weight- and input-gradient
 * products that read dX next.
 */
int cg_bn_bwd(float* dX, int lddx, const float* dY, int lddy, const float* X, int ldx,
              const float* Y, int ldy, const float* PM, int ldpm,
              const float* mean, const float* invstd, const float* gamma,
              float* dgamma, float* dbeta, int accumulate,
              float* gbar_out, float* gxbar_out, float slope, int M, int N,
              void* dxbf16, int lddxbf16, void* workspace, size_t workspace_bytes, void* stream);

/*
 * cg_bn_bwd on the `nseg` row segments of a cg_bn_fwd_seg call (mean/invstd
 * segment-major): dX per segment from that segment's statistics; dgamma/dbeta
 * (+)= the sum over all segments, added in segment order (what accumulating
 * .backward() calls of the separate batches give, GAN_functions.py:328-339).
 * gbar_out/gxbar_out: nseg*N values.
 */
int cg_bn_bwd_seg(float* dX, int lddx, const float* dY, int lddy, const float* X, int ldx,
                  const float* Y, int ldy, const float* PM, int ldpm,
                  const float* mean, const float* invstd, const float* gamma,
                  float* dgamma, float* dbeta, int accumulate,
                  float* gbar_out, float* gxbar_out, float slope, int seg_rows, int nseg, int N,
                  void* dxbf16, int lddxbf16, void* workspace, size_t workspace_bytes, void* stream);

/*
 * cg_bn_bwd whose incoming gradient dY is still in the n_slabs split-K partial sums that cg_gemm_slabs left in its
 * workspace (dense M x N slabs, one after the other): the input-gradient product of the layer above
 * (GAN_functions.py:89-94, 117-122 backward) and this BatchNorm backward then cost one pass over the gradient
 * instead of a slab reduce plus a pass.  The slabs are added in slab order, so dX, dgamma and dbeta equal, bit for
 * bit, what the slab reduce followed by cg_bn_bwd gives.  One-launch form only: cg_bn_bwd_slabs_ok(M, N) says
 * whether the shape qualifies (one segment of at most 512 rows, at least 256 columns).
 */
int cg_bn_bwd_slabs_ok(int M, int N);
int cg_bn_bwd_slabs(float* dX, int lddx, const float* slabs, int n_slabs, const float* X, int ldx,
                    const float* Y, int ldy, const float* PM, int ldpm, const float* mean, const float* invstd,
                    const float* gamma, float* dgamma, float* dbeta, int accumulate, float slope, int M, int N,
                    void* dxbf16, int lddxbf16, void* stream);

/*
 * BatchNorm backward-of-backward for the gradient penalty
 * (torch.autograd.grad(..., create_graph=True) then .backward(),
 * GAN_functions.py:199-203 through :166).  With u1 the first-order gradient
 * entering the BatchNorm backward, ubar = mean(u1), m = mean(u1*xhat) (from
 * cg_bn_bwd), and W the incoming second-order gradient w.r.t. u0:
 *   wbar = mean(W), r = mean(W*xhat), q = mean(W*u1)
 *   V_u1 = invstd*gamma*(W - wbar - xhat*r)
 *   V_h0 = -invstd^2*gamma*( m*(W-wbar) + r*(u1-ubar) + xhat*(q - wbar*ubar - 3*m*r) )
 *   dgamma (+)= M*invstd*(q - wbar*ubar - r*m)
 * vh_bf16 (may be NULL): bf16 image of V_h0 (leading dimension ldvh) for the weight-gradient product that reads it.
 * PM (may be NULL): V_u1 is stored multiplied by it elementwise - the activation/dropout factor of the layer before
 * the BatchNorm, which is all the chain does with V_u1 next (:167-168).
 */
int cg_bn_bwd_bwd(float* V_u1, int ldvu, float* V_h0, int ldvh, float* dgamma, int accumulate,
                  const float* W, int ldw, const float* U1, int ldu, const float* X, int ldx,
                  const float* mean, const float* invstd, const float* gamma,
                  const float* ubar, const float* mbar, int M, int N, void* vh_bf16, const float* PM, int ldpm,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- elementwise / gather ------------------------------------------------ */

/* out[i] = a*x[i] + b*y[i]  (y may be NULL); 2-D with leading dimensions. */
int cg_axpby(float* out, int ldo, float a, const float* x, int ldx, float b, const float* y, int ldy,
             int M, int N, void* stream);
/* out = x * y elementwise (2-D). */
int cg_mul(float* out, int ldo, const float* x, int ldx, const float* y, int ldy, int M, int N, void* stream);
/* out[m, :] = coef[idx]*scale for all elements (coef is a device array). */
int cg_fill_coef(float* out, int ldo, int M, int N, const float* coef, int idx, float scale, void* stream);
/* float32 -> bf16 (round to nearest even), 2-D. */
int cg_to_bf16(void* out, int ldo, const float* x, int ldx, int M, int N, void* stream);
/* out <- x together with its bf16 image out_bf16 (same leading dimension ldo): loading a batch into a workspace
 * whose image the bf16 products read (the critic's input rows in the generator step, GAN_functions.py:347,366-371). */
int cg_copy_bf16(float* out, void* out_bf16, int ldo, const float* x, int ldx, int M, int N, void* stream);

/* Embedding lookup (GAN_functions.py:135,181): out[m, 0:E] = table[labels[m*lstride], 0:E]; a negative label (a row
 * that belongs to no cell) gives zeros.
 * out_bf16 (may be NULL): bf16 image of `out`, same leading dimension. */
int cg_embed_gather(float* out, int ldo, const float* table, int E, const int64_t* labels, int lstride,
                    int M, int n_types, void* out_bf16, void* stream);
/*
 * Sums of the rows of X (M x N) by label: out[t, n] (+)= scale * sum over rows m with labels[m*lstride] == t of
 * X[m, n], out dense (n_types x N); rows with a label outside [0, n_types) are skipped; deterministic order;
 * n_types <= 32.  The gradient of an embedding table that enters a Linear layer beside the data
 * (GAN_functions.py:161-165,181-184) is  sum over rows with label t of (dh W_e)[m]  =  (sum over those rows of
 * dh[m]) W_e:  summing the hidden gradients by label first turns a rows x E x hidden product and a scatter into
 * this pass and an n_types x E x hidden product.  minus_labels (may be NULL): a second label per row at which the row
 * is subtracted in the same pass (the difference of two such sums over one X).
 */
int cg_label_colsum(float* out, const float* X, int ldx, const int64_t* labels, int lstride, int M, int N,
                    int n_types, float scale, int accumulate, const int64_t* minus_labels, int mstride, void* stream);
/* Its gradient: dtable[t, e] (+)= sum over rows with label t of dout[m, e] (deterministic order). */
int cg_embed_scatter(float* dtable, int E, const float* dout, int lddo, const int64_t* labels, int lstride,
                     int M, int n_types, int accumulate, void* stream);

/*
 * reparameterize (GAN_functions.py:127-130): z = mu + eps*exp(0.5*log_var).
 * mu and log_var are the two halves [0:L) and [L:2L) of each row of `ml`
 * (row stride ldml).
 */
int cg_reparam_fwd(float* z, int ldz, const float* ml, int ldml, const float* eps, int M, int L, void* stream);
/*
 * Gradient into [dmu | dlog_var] (row stride ldd): from dz through the
 * reparameterisation plus the KL term of the generator loss
 * (GAN_functions.py:375-376): kl = -0.5*mean(1 + lv - mu^2 - exp(lv)) scaled by
 * coef[kl_idx]*kl_weight.
 */
int cg_reparam_bwd(float* dml, int ldd, const float* dz, int lddz, const float* ml, int ldml, const float* eps,
                   const float* coef, int kl_idx, float kl_weight, int M, int L, void* stream);

/* normalize_to_library_size (GAN_functions.py:152-155): y = relu(x)*lib/(sum_row relu(x)+1e-10); rowsum kept. */
int cg_libsize_fwd(float* Y, int ldy, const float* X, int ldx, float* rowsum, float library_size,
                   int M, int N, void* stream);
/* Its backward: dX = [x>0] * s*(dY - sum_row(dY*relu(x))/ (rowsum+1e-10)), s = lib/(rowsum+1e-10). */
int cg_libsize_bwd(float* dX, int lddx, const float* dY, int lddy, const float* X, int ldx, const float* rowsum,
                   float library_size, int M, int N, void* stream);

/* ---- gradient penalty pieces (GAN_functions.py:194-204) -------------------- */
/* interpolates = alpha*real + (1-alpha)*fake, alpha one value per row (:195-196). */
int cg_gp_interp(float* out, int ldo, const float* real, int ldr, const float* fake, int ldf,
                 const float* alpha, int M, int G, void* stream);
/*
 * The three input batches of a critic step in one pass (:326-336): rows [0, M) of `out` <- real, [M, 2M) <- fake,
 * [2M, 3M) <- alpha*real + (1-alpha)*fake; lab_out (3M x n_cov int64, may be NULL) <- real_lab, fake_lab, fake_lab
 * (the critic sees the generated and the interpolated batch with the labels of the generated one, :328-329).
 * interp_block (>= 2): the interpolates go to rows [interp_block M, (interp_block + 1) M) instead of [2M, 3M) - a
 * caller may keep another row block between the generated batch and the interpolates (cg_gp_combine's caller does);
 * lab_out then has (interp_block + 1) M rows too, the rows of blocks in between set to -1 (no cell: cg_embed_scatter
 * skips such rows).
 * out_bf16 (may be NULL): bf16 image of `out`, same leading dimension.
 */
int cg_critic_inputs(float* out, int ldo, const float* real, int ldr, const float* fake, int ldf,
                     const float* alpha, int M, int G, int64_t* lab_out, const int64_t* real_lab,
                     const int64_t* fake_lab, int n_cov, int interp_block, void* out_bf16, void* stream);
/*
 * The critic's first Linear (GAN_functions.py:165, on [x | emb], :184) applied to the interpolated batch of the
 * gradient penalty (:196-199) without a product.  The layer is linear and the interpolates are
 * alpha x_real + (1 - alpha) x_fake with the generated batch's labels, so
 *   h_interp = alpha h_real + (1 - alpha) h_fake + alpha (emb_fake - emb_real) We^T
 * H (3M x N, leading dimension ldh) holds h_real in rows [0, M) and h_fake in [M, 2M), bias included, and receives
 * h_interp in rows [2M, 3M).  emb (may be NULL: no covariates): 2M x E embedding rows of the real and the generated
 * batch; We: the embedding columns of the layer's weight, stored [N][E] with leading dimension ldw.
 */
int cg_interp_hidden(float* H, int ldh, const float* alpha, const float* emb, int lde, int E,
                     const float* We, int ldw, int M, int N, void* stream);

/*
 * The penalty's first-layer weight-gradient term folded into the Wasserstein terms (:326-339 with :196-199).  The
 * interpolates are x_i = a x_r + (1 - a) x_f row by row, so with vh0 the second-order chain's gradient at the first
 * layer's output on the interpolates,  vh0^T x_i = (a vh0)^T x_r + ((1 - a) vh0)^T x_f:  the critic step's whole
 * gradient of the first layer's weight is  (dh0_r + a vh0)^T x_r + (dh0_f + (1 - a) vh0)^T x_f + u0^T V,  three row
 * blocks instead of four - one product the size of the first layer less per critic and step.  In place on D
 * (2M x N: dh0 of the real rows, then of the generated rows):  D[m] += a_m VH[m],  D[M + m] += (1 - a_m) VH[m].
 * The embedding columns of the interpolates are those of the generated batch, not interpolated; the product over
 * [x | emb] with the combined rows therefore misses  (a vh0)^T (emb_f - emb_r)  there: AV (M x N) receives a vh0 and
 * dE (M x E, emb = rows of embeddings of the real then the generated batch, may be NULL) receives emb_f - emb_r for
 * that correction product.  d_bf16 (may be NULL): bf16 image of D.
 */
int cg_gp_combine(float* D, int ldd, const float* VH, int ldv, const float* alpha, float* AV, int ldav,
                  const float* emb, int lde, int E, float* dE, int ldde, int M, int N, void* d_bf16, void* stream);

/*
 * Row norms and the penalty (:203): n_m = ||UX[m,:]||_2; rowpen[m] = (n_m-1)^2
 * (their mean is the penalty); V[m,:] = (lambda*2/M)*((n_m-1)/n_m)*UX[m,:] is the
 * gradient of lambda*penalty w.r.t. UX.  V may alias UX.  v_bf16 (may be NULL): bf16 image of V (leading
 * dimension ldv) for the two products that read V next.
 */
int cg_gp_penalty(float* V, int ldv, float* rowpen, const float* UX, int ldu, float lambda_gp,
                  int M, int G, void* v_bf16, void* stream);
/* cg_gp_penalty whose row terms are the partial sums of a reduction (see "reductions" below): parts[m] = (n_m-1)^2
 * and *nparts = M, M <= CG_MAX_PARTS - the penalty's mean (:203) without a reduction launch of its own. */
int cg_gp_penalty_parts(float* V, int ldv, float* parts, int* nparts, const float* UX, int ldu, float lambda_gp,
                        int M, int G, void* v_bf16, void* stream);

/* ---- reductions / losses ---------------------------------------------------
 * Each reduction writes at most CG_MAX_PARTS float partial sums to
 * `parts` and their count to *nparts (device int); cg_loss_finalize (or any
 * consumer) adds them in index order in float64, so results are deterministic.
 */
/* sum of all elements of X (critic means :330,:335,:347; row-penalty mean :203). */
int cg_reduce_sum(float* parts, int* nparts, const float* X, int ldx, int M, int N, void* stream);
/* nseg cg_reduce_sum in one launch: row segment k (seg_rows rows) of X into reduction k of nseg consecutive ones
 * (parts + k * CG_MAX_PARTS, nparts + k): the critic's means over the real and the generated batch (:330,:335) when
 * both were evaluated as one stacked batch.  The same partial sums as nseg calls. */
int cg_reduce_sum_seg(float* parts, int* nparts, const float* X, int ldx, int seg_rows, int nseg, int N, void* stream);
/* sum |A - B| (L1 cycle loss :353-357). */
int cg_reduce_l1(float* parts, int* nparts, const float* A, int lda, const float* B, int ldb, int M, int N, void* stream);
/* sum pred - target*log(pred + 1e-10): poisson_loss (GAN_functions.py:216-217) before its mean; the cycle term of
 * the hyper-parameter search on counts data (GAN_optimize_optuna.ipynb cell-1, train_epoch). */
int cg_reduce_poisson(float* parts, int* nparts, const float* pred, int ldp, const float* target, int ldt, int M, int N,
                      void* stream);
/* sum (A[m,:] - C[labels[m],:])^2 (cell-type centroid MSE :359-364); labels are clamped to [0, n_types). */
int cg_reduce_centroid_sq(float* parts, int* nparts, const float* A, int lda, const float* C, int ldc,
                          const int64_t* labels, int lstride, int M, int N, int n_types, void* stream);
/* sum of (1 + lv - mu^2 - exp(lv)) over [mu | lv] rows (KL :375-376). */
int cg_reduce_kl(float* parts, int* nparts, const float* ml, int ldml, int M, int L, void* stream);
/* out[n] (+)= scale * sum_m X[m,n]: column means (feature matching :368,:372; scale = 1/M) and the
 * bias gradients of nn.Linear (scale = 1). */
int cg_colsum(float* out, const float* X, int ldx, int M, int N, float scale, int accumulate, void* stream);
/* Up to four contiguous device-to-device copies in one launch (a job with bytes = 0 is skipped; addresses and sizes
 * multiples of 4): the batches and labels of one training iteration (GAN_functions.py:306-319, the .to(device) of the
 * loop body) into the fixed buffers that a captured step reads. */
int cg_copy4(void* dst0, const void* src0, long long bytes0, void* dst1, const void* src1, long long bytes1,
             void* dst2, const void* src2, long long bytes2, void* dst3, const void* src3, long long bytes3,
             void* stream);
/* Up to three cg_colsum in one launch (the bias gradients of consecutive nn.Linear layers at the end of a backward
 * chain, GAN_functions.py:165-174); a job whose out is NULL is skipped.  The same bits as three calls. */
int cg_colsum3(float* out0, const float* X0, int ldx0, int M0, int N0, float scale0, int accumulate0,
               float* out1, const float* X1, int ldx1, int M1, int N1, float scale1, int accumulate1,
               float* out2, const float* X2, int ldx2, int M2, int N2, float scale2, int accumulate2, void* stream);
/* sum (A - B)^2 (feature-matching MSE between two mean vectors :368,:372). */
int cg_reduce_sqdiff(float* parts, int* nparts, const float* A, int lda, const float* B, int ldb, int M, int N,
                     void* stream);
/* mean over columns of the unbiased column variance (metrics :418-419), one float;
 * strip_ws: (N+31)/32 doubles of caller workspace. */
int cg_colvar_mean(float* out, double* strip_ws, const float* X, int ldx, int M, int N, void* stream);

/* dA = coef[idx]*scale*sign(A-B) (gradient of the L1 term); dA may alias A. */
int cg_l1_bwd(float* dA, int ldd, const float* A, int lda, const float* B, int ldb,
              const float* coef, int idx, float scale, int M, int N, int accumulate, void* stream);
/* dA (+)= coef[idx]*scale*(1 - target/(pred + 1e-10)): gradient of poisson_loss (GAN_functions.py:216-217) w.r.t. pred. */
int cg_poisson_bwd(float* dA, int ldd, const float* pred, int ldp, const float* target, int ldt,
                   const float* coef, int idx, float scale, int M, int N, int accumulate, void* stream);
/* dA (+)= coef[idx]*scale*(A[m,:] - C[labels[m],:]) (gradient of the centroid MSE). */
int cg_centroid_bwd(float* dA, int ldd, const float* A, int lda, const float* C, int ldc,
                    const int64_t* labels, int lstride, const float* coef, int idx, float scale,
                    int M, int N, int n_types, int accumulate, void* stream);
/* dF[m,n] (+)= coef[idx]*scale*(mf[n]-mr[n]) for every row (gradient of the feature-matching MSE). */
int cg_fm_bwd(float* dF, int ldd, const float* mean_fake, const float* mean_real,
              const float* coef, int idx, float scale, int M, int N, int accumulate, void* stream);

/*
 * Loss bookkeeping of one generator step, on the device in float64 so that no
 * host round trip sits between the forward and the backward
 * (GAN_functions.py:347-406).  Inputs are the partial sums written by the
 * reductions above, laid out as parts[slot*CG_MAX_PARTS + i] with counts
 * nparts[slot]; slot meaning and normalisation are given by the caller in
 * `desc` (see cgrna_amd/engine.py).  Updates the five moving averages
 * (alpha = 0.3), computes the scales, and writes
 *   losses_out: 12 doubles - g_adv, g_cycle, g_celltype, g_fm, kl, g_loss, the five scales, the sixth term
 *   coef_out:   6 floats - lambda_i*scale_i for the backward kernels (the sixth: lambda_6)
 * A sixth term with a fixed weight, outside the moving averages and the re-scaling, is an extension: the identity
 * loss of CycleGAN, which GAN_functions.py does not have (n_terms[5] = 0 and lambda[5] = 0 give the reference's
 * g_loss).
 */
typedef struct {
  int n_terms[6];        /* number of (slot, weight) pairs per loss: adv, cycle, celltype, fm, kl, identity */
  int slot[6][4];
  double weight[6][4];   /* loss_k = sum_j weight[k][j] * total(slot[k][j]) */
  double lambda[6];
  double ema_alpha;
  const float* weights;  /* NULL, or 5 device floats used as coef_out as they are: no moving averages and no
                            re-scaling (the per-epoch curriculum weights of GAN_optimize_optuna.ipynb cell-1,
                            get_curriculum_weights x the epoch-level scales of train_epoch) */
} cg_gloss_desc;
int cg_gloss_finalize(double* ema_state, double* losses_out, float* coef_out, const float* parts,
                      const int* nparts, const cg_gloss_desc* desc, void* stream);
/* totals[i] = scale[i] * sum(parts[slot i]) for n slots (float64), generic. */
int cg_sum_parts(double* totals, const float* parts, const int* nparts, const int* slots, const double* scales,
                 int n, void* stream);

/*
 * ResidualBlock of the latent stage on the token view of the latent
 * (GAN_functions.py:54-68: x + BN(Linear(ReLU(BN(Linear(x))))) with Linear(16,16),
 * applied to the (batch*16, 16) token matrix at :129-132).  All matrices are dense
 * (n_tok, 16) float32, 16-byte aligned; W is [out][in] like nn.Linear.weight.
 * Column statistics travel between launches as per-workgroup float64 partial
 * sums: `stat_*` buffers hold cg_tok_blocks(n_tok)*32 doubles, `wpart*` buffers
 * cg_tok_blocks(n_tok)*256 floats.  Train mode only (batch statistics).
 *   forward:
 *     cg_tok_lin_fwd      T1 = X W1^T + b1;  stat_out <- sums of T1, T1^2
 *     cg_tok_bn_lin_fwd   R = relu(bn1(T1)); T2 = R W2^T + b2; stat_out <- sums of T2, T2^2;
 *                         mean_out/invstd_out (16) saved, running stats updated (momentum, unbiased var)
 *     cg_tok_bn_res_fwd   Y = X + bn2(T2); mean/invstd saved, running stats updated
 *   backward (autograd of the same lines), given dY = dL/dY:
 *     cg_tok_bn_bwd_stats stat_out <- sums of dY, dY*xhat2
 *     cg_tok_bwd_mid      dT2 = bn2'(dY); wpart <- partial dT2^T R; dgamma2/dbeta2 += ;
 *                         G1 = (dT2 W2) * (R > 0); stat_out <- sums of G1, G1*xhat1
 *     cg_tok_bwd_in       dT1 = bn1'(G1); wpart <- partial dT1^T X; dgamma1/dbeta1 += ;
 *                         dX = dY + dT1 W1
 *     cg_tok_wgrad_reduce dW2 += sum of wpart2, dW1 += sum of wpart1 (workgroup order)
 * The Linear biases feed a BatchNorm, so their gradient is identically zero and is not produced.
 */
int cg_tok_blocks(int n_tok);
int cg_tok_lin_fwd(float* T1, double* stat_out, const float* X, const float* W, const float* b,
                   int n_tok, void* stream);
int cg_tok_bn_lin_fwd(float* R, float* T2, double* stat_out, const float* T1, const double* stat_in,
                      const float* gamma, const float* beta, float eps, float* mean_out,
                      float* invstd_out, float* running_mean, float* running_var, float momentum,
                      const float* W2, const float* b2, int n_tok, void* stream);
int cg_tok_bn_res_fwd(float* Y, const float* T2, const double* stat_in, const float* gamma,
                      const float* beta, float eps, float* mean_out, float* invstd_out,
                      float* running_mean, float* running_var, float momentum, const float* X,
                      int n_tok, void* stream);
int cg_tok_bn_bwd_stats(double* stat_out, const float* dY, const float* T2, const float* mean,
                        const float* invstd, int n_tok, void* stream);
int cg_tok_bwd_mid(float* G1, double* stat_out, float* wpart, float* dgamma2, float* dbeta2,
                   const float* dY, const float* T2, const float* R, const float* T1,
                   const double* stat_in, const float* mean2, const float* invstd2,
                   const float* gamma2, const float* W2, const float* mean1, const float* invstd1,
                   int n_tok, void* stream);
int cg_tok_bwd_in(float* dX, float* wpart, float* dgamma1, float* dbeta1, const float* G1,
                  const float* T1, const float* X, const float* dY, const double* stat_in,
                  const float* mean1, const float* invstd1, const float* gamma1, const float* W1,
                  int n_tok, void* stream);
int cg_tok_wgrad_reduce(float* dW2, float* dW1, const float* wpart2, const float* wpart1, int n_tok,
                        void* stream);

/*
 * Fused multi-tensor Adam with amsgrad (torch.optim.Adam(..., betas=(0.5,0.9),
 * amsgrad=True), GAN_functions.py:223-225, stepped at :339-340,:409) over one
 * flat parameter arena: p, g, m, v, vmax are arrays of n floats.
 *   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; vmax = max(vmax, v);
 *   p -= (lr/(1-b1^t)) * m / (sqrt(vmax)/sqrt(1-b2^t) + eps)
 * lr_and_step: device array {lr, step t} as two doubles (so captured graphs can
 * be replayed with a new lr / step).  p_bf16 (may be NULL) receives the updated
 * parameters rounded to bf16.  vmax NULL: no running maximum, the denominator
 * uses v (torch.optim.Adam's default, as GAN_optimize_optuna.ipynb cell-1,
 * train_with_curriculum builds its optimizers).
 */
int cg_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, size_t n,
                    const double* lr_and_step, float beta1, float beta2, float eps,
                    void* p_bf16, void* stream);

/*
 * Fused token self-attention + LayerNorm of the generator bottleneck
 * (GeneSelfAttention.forward, GAN_functions.py:34-36, with
 * nn.MultiheadAttention(E, heads, dropout=0.1, batch_first=True) on its
 * need_weights=True path; E = latent_dim / 16 and heads come from the config,
 * GAN_functions.py:82,:103-113).  Per cell X is (16 tokens, E features):
 *   qkv = X Win^T + bin; A = softmax(q k^T / sqrt(E / heads)); P = A * keep;
 *   o = concat_heads(P v) Wout^T + bout;  Y = LayerNorm(X + o)   (eps 1e-5)
 * keep (NULL in eval mode) is the dropout keep-mask already divided by 0.9,
 * laid out (cell*heads + head, 16, 16).  `params` is one block of
 * CG_ATTN_NPARAM_E(E) floats: Win[3E][E] | bin[3E] | Wout[E][E] | bout[E] |
 * ln_w[E] | ln_b[E] (the order of the module's state_dict).  E = 16 with 4
 * heads (the reference's default, GAN_main.py:52-53) has a kernel of its own,
 * one wave per cell with everything in registers; E in {4, 8, 16, 32} with any
 * head count dividing E runs a general kernel (one wave per workgroup, stages
 * exchanged through LDS).  Anything else returns CG_ERR_UNSUPPORTED.
 */
#define CG_ATTN_NPARAM_E(E) (4 * (E) * (E) + 6 * (E))
#define CG_ATTN_NPARAM CG_ATTN_NPARAM_E(16)
int cg_attn_ln_fwd(float* Y, const float* X, const float* keep, const float* params, int n_cells, int E, int heads,
                   void* stream);
/*
 * One residual-attention block of the bottleneck (GAN_functions.py:142-145: ResidualBlock :54-68, then
 * GeneSelfAttention :28-36) from the ResidualBlock's second BatchNorm on, in one launch, for E = 16 with 4 heads
 * (CG_ERR_UNSUPPORTED otherwise): what cg_tok_bn_res_fwd and cg_attn_ln_fwd do one after the other, and - when
 * T1_next is not NULL - cg_tok_lin_fwd of the NEXT block:
 *   Xres = Xin + bn2(T2)            (stat_in: the n_stat_in partial column sums of T2, T2^2 that cg_tok_bn_lin_fwd wrote;
 *                                    mean_out / invstd_out / running statistics as cg_tok_bn_res_fwd)
 *   Y = LayerNorm(Xres + attention(Xres))
 *   T1_next = Y W1_next^T + b1_next; stat_next <- cg_attn_block_blocks(n_cells) x 32 partial sums of T1_next, T1_next^2
 * cg_tok_bn_lin_fwd_from is cg_tok_bn_lin_fwd reading n_stat_in partials (here: cg_attn_block_blocks(n_cells)) instead
 * of cg_tok_blocks(n_tok).  Two launches per block instead of four; Xres and T1_next are the bits of the separate
 * launches, the statistics of T1_next differ from theirs in the order of additions only.
 */
/*
 * cg_attn_block_bwd: cg_attn_ln_bwd that also takes the first step of the ResidualBlock's backward below it
 * (cg_tok_bn_bwd_stats): with g = dX it writes cg_attn_bwd_blocks(n_cells) x 32 partial sums of g and
 * g * (T2 - mean2) * invstd2 to stat_out, which cg_tok_bwd_mid_from (cg_tok_bwd_mid reading n_stat_in partials) adds
 * up.  E = 16 with 4 heads only.  reduce_now = 0: the partial parameter gradients stay in wpart (dparams is not
 * touched) for cg_attn_param_reduce_blocks.
 */
int cg_attn_block_blocks(int n_cells);
int cg_attn_block_bwd(float* dX, float* dparams, float* wpart, const float* dY, const float* X, const float* keep,
                      const float* params, int n_cells, int E, int heads, int accumulate, const float* T2,
                      const float* mean2, const float* invstd2, double* stat_out, int reduce_now, void* stream);
/* The parameter-gradient reduces of several residual-attention blocks in one launch each, for callers that kept the
 * blocks' per-workgroup partial sums apart (cg_attn_block_bwd with reduce_now = 0 and its own wpart per block;
 * cg_tok_bwd_mid / cg_tok_bwd_in with their own wpart per block): block k reads wpart + k * wstride and adds into
 * dparams + k * dstride (dW2 / dW1 + k * dstride), strides in floats - the blocks of a generator lie at equal
 * distances in a flat parameter arena.  The same sums as cg_attn_ln_bwd's reduce / cg_tok_wgrad_reduce per block. */
int cg_attn_param_reduce_blocks(float* dparams, long long dstride, const float* wpart, long long wstride,
                                int n_param_blocks, int n_cells, int accumulate, void* stream);
int cg_tok_wgrad_reduce_blocks(float* dW2, float* dW1, long long dstride, const float* wpart2, const float* wpart1,
                               long long wstride, int n_res_blocks, int n_tok, void* stream);
int cg_tok_bwd_mid_from(float* G1, double* stat_out, float* wpart, float* dgamma2, float* dbeta2, const float* dY,
                        const float* T2, const float* R, const float* T1, const double* stat_in, int n_stat_in,
                        const float* mean2, const float* invstd2, const float* gamma2, const float* W2,
                        const float* mean1, const float* invstd1, int n_tok, void* stream);
int cg_attn_block_fwd(float* Y, float* Xres, const float* T2, const float* Xin, const double* stat_in, int n_stat_in,
                      const float* gamma, const float* beta, float eps, float* mean_out, float* invstd_out,
                      float* running_mean, float* running_var, float momentum, const float* keep, const float* params,
                      int n_cells, int E, int heads, float* T1_next, double* stat_next, const float* W1_next,
                      const float* b1_next, void* stream);
int cg_tok_bn_lin_fwd_from(float* R, float* T2, double* stat_out, const float* T1, const double* stat_in, int n_stat_in,
                           const float* gamma, const float* beta, float eps, float* mean_out, float* invstd_out,
                           float* running_mean, float* running_var, float momentum, const float* W2, const float* b2,
                           int n_tok, void* stream);
/*
 * Backward of cg_attn_ln_fwd; the forward is recomputed from X and keep, so no
 * activations are saved.  dX receives the input gradient; parameter gradients
 * are summed per workgroup into `wpart` (cg_attn_bwd_ws(n_cells, E, heads)
 * floats of caller workspace; -1 for a geometry that is not built) and then in
 * workgroup order into dparams (same layout as params), `accumulate` adding to
 * what is there.  cg_attn_bwd_blocks(n_cells) * CG_ATTN_NPARAM is the same
 * figure for E = 16, heads = 4.
 */
int cg_attn_bwd_blocks(int n_cells);
long long cg_attn_bwd_ws(int n_cells, int E, int heads);
int cg_attn_ln_bwd(float* dX, float* dparams, float* wpart, const float* dY, const float* X, const float* keep,
                   const float* params, int n_cells, int E, int heads, int accumulate, void* stream);

/*
 * Measurement aid; no counterpart in the reference.  Between cg_gemm_timing_begin(max_calls) and cg_gemm_timing_end
 * every cg_gemm / cg_gemm_ex / cg_gemm_slabs call made by the calling thread (at most max_calls of them) carries a
 * pair of HIP events attached to its own dispatches: the start of its first kernel and the end of its last one (the
 * split-K slab reduce, when the call launches one).  cg_gemm_timing_end waits for the device, writes the elapsed
 * milliseconds of the first `capacity` calls, in call order, to ms (-1 for a call that launched nothing), destroys
 * the events and returns the number of calls recorded (negative on error).  Not for use during stream capture.
 */
int cg_gemm_timing_begin(int max_calls);
int cg_gemm_timing_end(float* ms, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* CGRNA_H */
