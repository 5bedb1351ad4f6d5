/*
 * ctn.h — C ABI of libctn_hip.so: gfx950 (MI355X) kernels for the Conv-TasNet
 * training hot path.
 *
 * The reference (SaisumanaP/speech) has no FFI of its own for this path: every
 * operation is a stock torch.nn call.  Each entry point below therefore cites the
 * reference Python call site(s) whose arithmetic it replaces (paths relative to
 * the reference root).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer to fp32 unless typed otherwise.
 *  - Activations are (batch, C, ld) row-major with `T` valid time steps per row,
 *    ld >= T, ld % 4 == 0, rows 16-byte aligned.  Columns [T, ld) are padding:
 *    never read as data, may be overwritten with garbage.
 *  - The caller owns every buffer (inputs, outputs, workspaces, accumulators).
 *    The library never allocates, frees or synchronises; every call only enqueues
 *    kernels on `stream` (a hipStream_t).
 *  - Accumulator outputs (documented "+=") must be zeroed by the caller.
 *  - Return value: 0 on success, negative CTN_ERR_* otherwise;
 *    ctn_last_error() gives the message of the calling thread's last failure.
 *
 * Layer-norm statistics ("stat" arrays).  Both norms are expressed as
 *      n = (u - mean) * rstd * gamma[c] + beta[c]
 *  with mean/rstd indexed [b * sb + t * st]:
 *      gLN (GlobalLayerNorm, src/modules/norm.py:11-29): one value per sample, sb = 1, st = 0,
 *          rstd = 1/sqrt(var + eps);
 *      cLN (CumulativeLayerNorm1d, src/modules/norm.py:42-95): one value per frame, sb = ld_t, st = 1,
 *          rstd = 1/(sqrt(var) + eps)   (eps outside the root, norm.py:90).
 *  The backward of either norm is
 *      du = g * rstd + c1 + (u - mean) * c2,   g = gamma[c] * dn,
 *  with c1/c2 indexed like mean/rstd (ctn_gln_bwd_coefs / ctn_cln_bwd_coefs).
 */
#ifndef CTN_H
#define CTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scalar fp64 accumulators are replicated CTN_STAT_SLOTS ways to avoid same-address atomic contention:
 *   norm sums   double[batch][CTN_STAT_SLOTS][2]   (every `sums` / `esums` / `isums` argument below)
 *   dalpha      double[CTN_STAT_SLOTS]             (every PReLU-slope gradient accumulator)
 * The value is the sum over the slot axis; ctn_gln_finalize / ctn_gln_bwd_coefs do that sum themselves. */
#define CTN_STAT_SLOTS 32

#define CTN_OK 0
#define CTN_ERR_ARG (-1)     /* bad shape / alignment / null pointer */
#define CTN_ERR_LAUNCH (-2)  /* hipLaunchKernel reported an error */

/* operand transform flags (GEMM operands, norm kernels) */
#define CTN_XF_PRELU 1    /* u = x > 0 ? x : alpha * x            (nn.PReLU, src/models/tdcn.py:119,184) */
#define CTN_XF_NORM 2     /* n = (u - mean) * rstd * gamma + beta (src/modules/norm.py:27,90) */
#define CTN_XF_NORMBWD 4  /* dy = (gamma*X*rstd + c1 + (prelu(X2)-mean)*c2) * (X2 > 0 ? 1 : alpha) */

/* GEMM epilogue modes */
#define CTN_EPI_STORE 0      /* Y = acc + bias */
#define CTN_EPI_STATS 1      /* STORE, plus sums of prelu(Y, ealpha) and its square: esums (gLN) or ecol (cLN) */
#define CTN_EPI_HEADS 2      /* rows < Mo: Y = acc + bias + R ; rows >= Mo: Y2 (+)= acc + bias2 */
#define CTN_EPI_SIGMOID 3    /* Y = sigmoid(acc + bias) */
#define CTN_EPI_PRELUBWD 4   /* Y = acc * (E > 0 ? 1 : ealpha); esums[0] += sum acc * min(E, 0) */
#define CTN_EPI_NORMBWD 5    /* Y = acc; norm-backward sums of g = egamma*acc and g*xhat(E): esums / ecol */
#define CTN_EPI_ADD 6        /* Y = acc + bias + (R ? R : 0) */

const char* ctn_version(void);
const char* ctn_last_error(void);
/* sizeof() of the argument structs below, for bindings to check their mirror of this header */
int ctn_sizeof_gemm_nn_args(void);
int ctn_sizeof_gemm_nt_args(void);
int ctn_sizeof_dw_fwd_args(void);
int ctn_sizeof_dw_bwd_args(void);

/* Arithmetic of the products of ctn_gemm_nn / ctn_gemm_nt (field `arith` of their argument structs).  The library
 * keeps no mode of its own: every call says what it wants, so calls with different modes can be in flight together and
 * a captured graph holds exactly what its calls asked for.
 *   CTN_ARITH_DEFAULT (0): the same as CTN_ARITH_BF16X3.
 *   CTN_ARITH_F32:    v_mfma_f32_32x32x2_f32 on the fp32 operands.
 *   CTN_ARITH_BF16X3: every fp32 operand is split into three bf16 pieces, x = hi + mid + lo EXACTLY (each piece the next
 *       eight significant bits), and the six products of pieces above 2^-22 of the largest run on
 *       v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  No operand is rounded; what is left out (mid*lo, lo*mid, lo*lo)
 *       is of the size of one fp32 rounding of the product.  Meets the same bounds as CTN_ARITH_F32 in every test.
 *   CTN_ARITH_BF16X2: x = hi + lo (17 mantissa bits, pieces rounded to nearest) and three products.  Each GEMM then
 *       agrees with fp32 arithmetic to about 1e-5 of its largest output instead of 1e-6: a throughput mode for callers
 *       who accept that.
 * The register-staged kernels for small or unaligned shapes always use CTN_ARITH_F32. */
#define CTN_ARITH_DEFAULT 0
#define CTN_ARITH_F32 1
#define CTN_ARITH_BF16X2 2
#define CTN_ARITH_BF16X3 3

/* Which ctn_gemm_nn kernel consumes a prepared A operand (field `prep_kernel`; read only when A_prep is given).  Both
 * compute the products of CTN_ARITH_BF16X3 in its order: results are bit-identical to each other and to the call
 * without A_prep.
 *   CTN_PREP_DEFAULT (0): CTN_PREP_WAVE_SPLIT8 unless the call names a tile_width or carries the norm-backward operand
 *       transform (then CTN_PREP_WAVE_SPLIT).
 *   CTN_PREP_WAVE_SPLIT: the four waves of a block sit side by side along time and each splits its own fragment of the
 *       fp32 activation tile in registers; no activation element is split twice in a block, nothing is written back to LDS.
 *   CTN_PREP_BLOCK_SPLIT: all threads of the block split the activation tile once into fragment order in LDS, one
 *       k-tile ahead of the matrix instructions (three landing buffers; more LDS, fewer resident blocks). */
#define CTN_PREP_DEFAULT 0
#define CTN_PREP_WAVE_SPLIT 1
#define CTN_PREP_BLOCK_SPLIT 2
/*   CTN_PREP_WAVE_SPLIT8: the wave-split kernel with eight waves over a 128 x 256 tile (tile_width is not read). */
#define CTN_PREP_WAVE_SPLIT8 3

/* Measurement aids without a counterpart in the reference: the rates the GPU itself sustains, for the benchmark to
 * report beside the kernels' own.
 * ctn_probe_copy: y = a (b NULL) or y = a + b, n floats (n % 4 == 0, 16-byte aligned), non-temporal 16-byte accesses;
 *   moves 8n (copy) or 12n (two reads, one write) bytes.
 * ctn_probe_mfma_f32: `blocks` blocks of four waves each issue iters x 4 v_mfma_f32_32x32x2_f32 back to back on operands
 *   held in registers (read once from in[0..65535]); out receives blocks*256 floats.  FLOPs = blocks*4*iters*4*4096. */
int ctn_probe_copy(const float* a, const float* b, float* y, int64_t n, void* stream);
/* y = a with plain (temporal) 16-byte accesses, 16 KiB chunks in ascending (reverse = 0) or descending block order: measures
 * what reading a producer's output last-written-first gains from the die-level cache. */
int ctn_probe_stream(const float* a, float* y, int64_t n, int reverse, void* stream);
int ctn_probe_mfma_f32(const float* in, float* out, int blocks, int iters, void* stream);
/* the same with v_mfma_f32_32x32x16_bf16 on bf16 bit patterns drawn from `in`: FLOPs = blocks*4*iters*4*32768 */
int ctn_probe_mfma_bf16(const float* in, float* out, int blocks, int iters, void* stream);

/* dst[k * ldd + coff + m] = src[m * K + k]  (m < M, k < K).  Lays a Conv1d(kernel_size=1) weight
 * (M_out, K_in) out k-major for ctn_gemm_nn. */
int ctn_transpose(const float* src, int M, int K, float* dst, int64_t ldd, int64_t coff, void* stream);
/* The same for any number of matrices in one launch per 64 of them (a training step lays out all its weights at once).
 * `items` is host memory, read during the call. */
typedef struct {
    const float* src; float* dst;
    int64_t ldd, coff;
    int32_t M, K;
} ctn_transpose_item;
int ctn_transpose_batch(const ctn_transpose_item* items, int n_items, void* stream);
int ctn_sizeof_transpose_item(void);

/* Prepared weights.  In CTN_ARITH_BF16X3 the A operand of ctn_gemm_nn (a weight matrix, read by every block of the
 * launch) can be split into its three bf16 pieces ONCE per optimizer step instead of by every wave that reads it, and
 * laid out the way the matrix instruction takes it.  A prepared K x M matrix (A[k][m], k the contraction index) is
 *   bf16 planes[3][Kp / 8][Mp][8],  Kp = K rounded up to 16, Mp = M rounded up to 128  (ctn_weight_prep_bytes(K, M)),
 * planes[p][g][m][q] = piece p (0 = leading eight bits, 1, 2 = the following eight each) of A[8 g + q][m], zero beyond
 * K and M: the caller zero-fills the buffer once, ctn_weight_prep writes only what its items cover.
 * An item copies one fp32 source matrix (rows x cols, row pitch ld) into the prepared matrix `dst` (logical size K x M)
 * at (k_off, m_off): transpose = 0 takes source rows as k (A[k_off + r][m_off + c] = src[r][c]; a Conv1d weight
 * (out, in) used for the input gradient), transpose = 1 takes source rows as m (A[k_off + c][m_off + r] = src[r][c];
 * the same weight used in the forward pass).  Several items may fill one prepared matrix (the two heads of a residual
 * layer side by side).  k_off % 8 == 0; dst 16-byte aligned.  `items` is host memory, read during the call; any number
 * of items, a launch per 56 of them.  Replaces ctn_transpose for callers in that arithmetic mode; no counterpart in the
 * reference (nn.Conv1d reads its weight where it is). */
typedef struct {
    const float* src; int64_t ld;
    void* dst;
    int32_t rows, cols;
    int32_t transpose;
    int32_t k_off, m_off;
    int32_t K, M;                     /* logical size of the prepared matrix dst belongs to */
    int32_t pad0;
} ctn_wprep_item;
int64_t ctn_weight_prep_bytes(int K, int M);
int ctn_weight_prep(const ctn_wprep_item* items, int n_items, void* stream);
int ctn_sizeof_wprep_item(void);

/* Chunking for dual-path separators and its adjoint.
 * ctn_segment: y[b][c][s][k] = xpad[b][c][s*P + k] with y dense (batch, C, S, K) and xpad = x (rows of T frames, pitch
 * ldx, batch stride xbs) zero-padded by pad_left frames in front and as far as the S chunks reach behind.  Replaces
 * Segment1d.forward (src/models/transform.py:15-31: F.unfold + permute) together with the F.pad in front of it
 * (src/models/dprnn_tasnet.py:335-341); pad_left = 0 and S = (T-K)/P+1 give Segment1d alone.
 * ctn_overlap_add: x[b][c][t] = sum over s of y[b][c][s][t + pad_left - s*P] for t < T.  Replaces OverlapAdd1d.forward
 * (transform.py:45-61: permute + F.fold) together with the trimming F.pad behind it (dprnn_tasnet.py:343-344).
 * Each is the other's backward. */
int ctn_segment(const float* x, int64_t xbs, int64_t ldx, int batch, int C, int T, int pad_left, int K, int P, int S,
                float* y, void* stream);
int ctn_overlap_add(const float* y, int batch, int C, int S, int K, int P, int pad_left, int T, float* x, int64_t xbs,
                    int64_t ldx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Axis shuffles of the dual-path blocks (src/models/dprnn.py:83,88 inside chunks; :135,140,144-146 across chunks:
 * permute(...).contiguous() between (batch, features, S, K) and the recurrences' (sequences, steps, features) rows, and
 * the residual sum behind the last of them), as batched strided 2-D transposes of fp32:
 *   dst[b2][b1][k][m] = src[b2][b1][m][k] (+ addend[b2][b1][k][m] when addend is not null)
 * for m < M, k < K, b1 < nb1, b2 < nb2; source rows are lds floats apart and destination rows ldd, sb1/sb2 and db1/db2
 * are the batch strides on the two sides; addend and dst share a layout and may be the same buffer.  nb1 * nb2 <= 65535.
 * ------------------------------------------------------------------------------------------- */
int ctn_permute(const float* src, float* dst, const float* addend, int M, int K, int64_t lds, int64_t ldd, int nb1,
                int64_t sb1, int64_t db1, int nb2, int64_t sb2, int64_t db2, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LSTM recurrence over whole sequences in one launch: the nn.LSTM(batch_first=True) of the dual-path blocks
 * (src/models/dprnn.py:60,85 inside chunks, :104-114,137 across chunks; gate order i, f, g, o as in torch), one or two
 * directions, zero initial state.  The caller hoists the input projection out of the recurrence:
 *   xproj  (N, T, D, 4H)  W_ih x_t + b_ih + b_hh for every sequence n, step t, direction d (one GEMM)
 *   whh    (D, 4H, H)     recurrent weights as torch stores them (weight_hh_l0, weight_hh_l0_reverse)
 *   hout   (N, T, D, H)   h_t; direction 1 walks t = T-1 .. 0 and stores its state at the step it has just consumed
 *   save   (N, T, D, 5, H) i, f, g, o after their non-linearities and the cell state c_t, for ctn_lstm_bwd
 * ctn_lstm_bwd walks the steps in the opposite order: from dhout (N, T, D, H) = dL/dh_t and `save` it writes
 *   dgates (N, T, D, 4H)  dL/d(gate pre-activations) = dL/dxproj,
 *   dbias  (D, 4H)        its sum over (N, T) = dL/d(b_ih) = dL/d(b_hh); may be null (not formed); cleared by the call,
 *                         accumulated with float atomics (summation order varies from run to run)
 * from which the weight and input gradients are plain GEMMs over (N, T).
 * H <= 128 and H % 4 == 0; all tensors dense and 16-byte aligned; fp32 in and out.  The recurrent products run on the
 * bf16 matrix instruction in the exact three-piece split of CTN_ARITH_BF16X3 (fp32 accumulation); the environment
 * variable CTN_LSTM_F32=1, read once, selects the f32 matrix instruction instead.
 * ------------------------------------------------------------------------------------------- */
int ctn_lstm_fwd(const float* xproj, const float* whh, int N, int T, int H, int D, float* hout, float* save,
                 void* stream);
int ctn_lstm_bwd(const float* dhout, const float* save, const float* whh, int N, int T, int H, int D, float* dgates,
                 float* dbias, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense products over feature-last rows: the input projection of the dual-path blocks' LSTMs (nn.LSTM's W_ih x_t + b_ih +
 * b_hh, src/models/dprnn.py:60-64,85 inside chunks, :104-114,137 across chunks), their linear layers (nn.Linear,
 * dprnn.py:62,87 and :116,139) and the weight / input gradients of both.  X, Y, dY are (R, features) fp32 row-major
 * with row strides ldx / ldy / lddy (in floats; views with a column offset are fine), R = sequences x steps.
 *   ctn_rows_gemm   Y[r][m] = sum_k X[r][k] * Wkm (+ bias[m]);  w_is_km = 0: W is (M, K) row-major, torch's (out, in)
 *                   layout, i.e. Y = X W^T + b (forward);  w_is_km = 1: W is (K, M) row-major, i.e. Y = X W (the input
 *                   gradient dX = dY W with the forward weight as it lies).  bias may be null.
 *   ctn_rows_wgrad  dW[m][k] += sign * sum_r dY[r][m] * X[r][k] and, when dbias is not null,
 *                   dbias[m] += sign * sum_r dY[r][m]; dW (M, K) with row stride lddw and dbias are accumulated into
 *                   with float atomics (the caller zeroes them; summation order varies from run to run).  sign = -1
 *                   subtracts a product (the sequence-boundary terms of a shifted product).
 * arith: CTN_ARITH_DEFAULT / CTN_ARITH_BF16X3 = the exact three-piece split with six bf16 matrix instructions per
 * product, as in ctn_gemm_nn; CTN_ARITH_F32 = v_mfma_f32_32x32x2_f32.  In ctn_rows_gemm the split form applies to
 * w_is_km = 0 (both operands contiguous along the contraction; a caller that wants it for a (K, M) weight passes the
 * transposed copy) and w_is_km = 1 always runs f32 instructions; ctn_rows_wgrad under CTN_ARITH_DEFAULT runs the
 * split form where its whole-output kernel applies (M a multiple of 512, K a multiple of 64 up to 128, vector layout:
 * the recurrent layers' weight gradients) and f32 instructions elsewhere; explicit CTN_ARITH_BF16X3 takes the split
 * form for any M > 64.
 * Any R, K, M >= 1; 16-byte aligned bases and row strides that are multiples of 4 floats take the vector-load path,
 * anything else a scalar one.
 * ------------------------------------------------------------------------------------------- */
int ctn_rows_gemm(const float* X, int64_t ldx, const float* W, int64_t ldw, int w_is_km, const float* bias, int64_t R,
                  int K, int M, float* Y, int64_t ldy, int arith, void* stream);
int ctn_rows_wgrad(const float* dY, int64_t lddy, const float* X, int64_t ldx, int64_t R, int M, int K, float* dW,
                   int64_t lddw, float* dbias, float sign, int arith, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward pass of a residual layer's two pointwise heads in one launch: what ctn_gemm_nn (heads' input gradient with the
 * CTN_EPI_NORMBWD sums) and ctn_gemm_nt (heads' weight and bias gradients) compute in two, with y2 and dY read once.
 * Reference: the output / skip Conv1d(H, B, 1) pair of src/models/tdcn.py:172-175,190-194 behind the norm of :187 (gLN).
 *   dY   = [dY1 (M1 rows: gradient at the residual output; M1 = 0 for a layer without it); dY2 (M2 rows: at the skip output)]
 *   dn2[b][c][t]  = sum_m Wh[m][c] dY[b][m][t]                     (written; (batch, H, T), row pitch lddn2)
 *   esums[b][slot][0] += sum_{c,t} gamma[c] dn2,   [1] += sum_{c,t} gamma[c] dn2 xhat2,  xhat2 = (prelu(y2) - mean_b) rstd_b
 *   dW1[m][c] += sum_{b,t} dY1[b][m][t] n2[b][c][t],  dW2 likewise,  n2 = gamma xhat2 + beta       (float atomics)
 *   rowsum1[m] += sum_{b,t} dY1[b][m][t],  rowsum2 likewise          (the bias gradients; either may be null)
 * W_prep: the (M1 + M2) x H matrix A[k = m][c] = Wh[m][c] in prepared form (ctn_weight_prep).  H % 128 == 0,
 * M1 + M2 = 128 or 256 with both multiples of 32, T >= 4, gLN statistics (one mean / rstd per batch element, stride ssb);
 * arithmetic CTN_ARITH_BF16X3.  Accumulated outputs are zeroed by the caller.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, T, H;
    int32_t M1, M2;
    const float* dY1; int64_t lddy1, dy1bs;
    const float* dY2; int64_t lddy2, dy2bs;
    const float* y2;  int64_t ldy2, y2bs;
    const float* alpha;
    const float* mean; const float* rstd; int64_t ssb;
    const float* gamma; const float* beta;
    const void* W_prep;
    float* dn2; int64_t lddn2, dn2bs;
    double* esums;
    float* dW1; int64_t lddw1;
    float* dW2; int64_t lddw2;
    float* rowsum1; float* rowsum2;
    int32_t ldmin, pad0;              /* set by the library */
} ctn_heads_bwd_args;
int ctn_sizeof_heads_bwd_args(void);
int ctn_heads_bwd(const ctn_heads_bwd_args* args, void* stream);

/* Finishing a global layer norm's reduction inside the launch that produces its sums (field `fin` of the argument structs
 * below; it applies to that launch's `esums` / `isums`).  With `count` given, the block that adds the LAST partial sums
 * of a batch element also turns the element's sums into the two scalars the consumers read, so no launch of
 * ctn_gln_finalize / ctn_gln_bwd_coefs is needed behind the producer:
 *   CTN_FIN_STATS: out0 = mean = S1/n, out1 = rstd = 1/sqrt(S2/n - mean^2 + eps)          (what ctn_gln_finalize writes)
 *   CTN_FIN_COEFS: out0 = c1 = -rstd*S1/n, out1 = c2 = -rstd^2*S2/n, rstd read from `rstd` (what ctn_gln_bwd_coefs writes)
 * count: uint32[batch], ZERO before the launch (every block of a batch element increments its entry once, after its
 * additions have been performed); count NULL = off.  Per-sample (gLN) sums only.  No counterpart in the reference. */
#define CTN_FIN_STATS 0
#define CTN_FIN_COEFS 1
typedef struct {
    uint32_t* count;
    float* out0; float* out1;
    const float* rstd;
    double n;
    float eps;
    int32_t kind;
} ctn_finish;

/* ---------------------------------------------------------------------------------------------
 * Pointwise (1x1) convolution as a batched GEMM on f32 MFMA:  Y[b] = A^T-layout(W) . xform(X[b])
 * Replaces nn.Conv1d(kernel_size=1) at src/models/conv_tasnet.py:335,341 (:371,:374) and
 * src/models/tdcn.py:86,173,175 (:116,:190,:194), forward and input-gradient, with the PReLU /
 * layer-norm / residual / skip-sum / sigmoid neighbours of those calls fused into operand load
 * and epilogue.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t M, T, K, batch;           /* Y is (batch, M, T); contraction length K */
    /* A, k-major: element (k, m) at A[k*lda + m] for k < K1 else A2[(k-K1)*lda2 + m]. K1 = K if unsplit */
    const float* A;  int64_t lda;
    const float* A2; int64_t lda2;
    int32_t K1;
    int32_t xflags;                   /* CTN_XF_* applied to X on load */
    const float* bias;                /* [M] (rows < Mo in HEADS) or NULL */
    const float* bias2;               /* HEADS: [M - Mo] */
    /* X (batch, K, ldx): rows k < K1 from X, k >= K1 from Xb */
    const float* X;  int64_t ldx,  xbs;
    const float* Xb; int64_t ldxb, xbbs;
    const float* X2; int64_t ldx2, x2bs;   /* NORMBWD: pre-activation tensor, row k */
    const float* alpha;               /* device scalar (PRELU / NORMBWD) */
    const float* mean; const float* rstd; const float* c1; const float* c2;
    int64_t ssb, sst;                 /* stat index = b*ssb + t*sst */
    const float* gamma; const float* beta;   /* [K] */
    float* Y; int64_t ldy, ybs;
    int32_t emode;
    int32_t Mo;                       /* HEADS split row */
    const float* R; int64_t ldr, rbs; /* residual (HEADS rows < Mo, ADD) */
    float* Y2; int64_t ldy2, y2bs;    /* HEADS rows >= Mo */
    int32_t accumulate;               /* HEADS: Y2 += instead of = */
    int32_t tile_width;               /* output-tile columns of the LDS-DMA kernel: 0 = choose, 64 or 128 */
    int32_t arith;                    /* CTN_ARITH_* */
    int32_t prep_kernel;              /* with A_prep: CTN_PREP_* (which kernel consumes the prepared operand) */
    const float* ealpha;              /* STATS / PRELUBWD / NORMBWD slope (NULL = 1, i.e. no PReLU) */
    double* esums;                    /* [batch][CTN_STAT_SLOTS][2] (+=), PRELUBWD: [CTN_STAT_SLOTS] */
    float* ecol; int64_t ecol_bs, ecol_ld;   /* cLN: [batch][2][ecol_ld] column sums (+=) */
    const float* E; int64_t lde, ebs; /* epilogue input (batch, M, lde) */
    const float* emean; const float* erstd; int64_t essb, esst;
    const float* egamma;              /* [M] */
    /* A in prepared form (ctn_weight_prep below), or NULL.  When given, A / lda / A2 / lda2 are not read and the call
     * must satisfy: arith CTN_ARITH_DEFAULT or CTN_ARITH_BF16X3, M > 64, M % 4 == 0, K % 16 == 0, K1 % 16 == 0, K <= 4096
     * and the 16-byte alignments of the LDS-DMA kernel (it is refused otherwise, there is no slower path behind it). */
    const void* A_prep;
    ctn_finish fin;                   /* STATS / NORMBWD epilogue with esums: finish the reduction in this launch */
} ctn_gemm_nn_args;

int ctn_gemm_nn(const ctn_gemm_nn_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient of a 1x1 convolution:  C[m][n] += sum_b sum_t xformA(A)[b][m][t] * xformB(B)[b][n][t]
 * (the weight-gradient half of autograd for the nn.Conv1d calls listed above), split over
 * (batch, time chunks) with fp32 atomic accumulation.  Optional fused reductions of the A operand:
 * bias gradient (row sums) and the PReLU-slope gradient of the NORMBWD transform.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t M, N, T, batch;
    const float* A;  int64_t lda,  abs_;      /* rows m < M1 */
    const float* Ab; int64_t ldab, abbs;      /* rows m >= M1 */
    int32_t M1;
    int32_t aflags;                           /* 0 or CTN_XF_NORMBWD */
    const float* A2; int64_t lda2, a2bs;      /* NORMBWD pre-activation */
    const float* a_alpha;
    const float* a_mean; const float* a_rstd; const float* a_c1; const float* a_c2;
    int64_t a_ssb, a_sst;
    const float* a_gamma;
    float* a_rowsum;  /* [M1]     += sum_t A'[m][t]      (NULL to skip) */
    float* a_rowsum2; /* [M - M1] */
    double* a_dalpha; /* NORMBWD: [CTN_STAT_SLOTS] += sum du * min(y, 0)  (NULL to skip) */
    const float* B; int64_t ldb, bbs;
    int32_t bflags;                           /* CTN_XF_PRELU | CTN_XF_NORM */
    int32_t kchunk;                           /* time steps per block, multiple of 32 */
    int32_t arith;                            /* CTN_ARITH_* */
    int32_t pad0;
    const float* b_alpha;
    const float* b_mean; const float* b_rstd; int64_t b_ssb, b_sst;
    const float* b_gamma; const float* b_beta;
    float* C;  int64_t ldc;                   /* rows m < M1  (+=) */
    float* C2; int64_t ldc2;                  /* rows m >= M1 (+=) */
    /* Optional by-products per batch element, for ctn_heads_bwd_coefs below (all NULL to skip; the LDS-DMA kernel only:
     * M, N, M1 multiples of 16):
     *   a_rowsum_b[b][m]  += sum_t A'[b][m][t]                                   (float [batch][M], zeroed by the caller)
     *   wdot_sums[b][slot] += sum_{m,n} Wd[m][n] * (this block's partial of C_b)  (double [batch][CTN_STAT_SLOTS], zeroed)
     * with Wd = wdot (rows m < M1, pitch ldwd) and wdot2 (rows m >= M1, pitch ldwd2) in the layout of C / C2. */
    float* a_rowsum_b;
    const float* wdot;  int64_t ldwd;
    const float* wdot2; int64_t ldwd2;
    double* wdot_sums;
    /* Optional workspace of the whole-output kernels (below): ws_floats floats, 16-byte aligned, contents irrelevant
     * before and after the call, not shared with a call that may run at the same time.  ctn_gemm_nt_ws_floats says how
     * many floats a call would use; NULL or too small: the kernels add their partial outputs with float atomics. */
    float* ws; int64_t ws_floats;
} ctn_gemm_nt_args;

int ctn_gemm_nt(const ctn_gemm_nt_args* args, void* stream);

/* The whole-output kernels of ctn_gemm_nt: for the two weight-gradient shapes of a residual layer and the mask
 * convolution's (M = 256 against N a multiple of 256; M a multiple of 256 against N = 128, with or without the NORMBWD
 * operand), three-piece arithmetic, global norm
 * statistics and enough frames to give every CU four to eight stages, one 512-thread block per CU keeps a 256 x 256 or
 * 256 x 128 output in its accumulators and every operand element is fetched, transformed and split once.  With a
 * workspace (fields ws / ws_floats) the blocks store their partial outputs and a second launch adds them in a fixed
 * order -- C and the row sums then get the same bits on every run -- instead of 16.8 M float atomics.
 * CTN_NT_WHOLE in the environment (read at every call): 0 = tile kernels only, 2 = the NORMBWD form only, 3 = the other
 * form only.  No counterpart in the reference (autograd of nn.Conv1d, src/models/tdcn.py:118-120,172-175).
 *   ctn_gemm_nt_ws_floats:      floats of workspace the call described by args would use (0: it takes the tile kernels)
 *   ctn_gemm_nt_whole_launches: calls of this process that took the whole-output kernels (a diagnostic for tests and
 *                               benchmarks that must know which kernels they measured) */
int64_t ctn_gemm_nt_ws_floats(const ctn_gemm_nt_args* args);
int64_t ctn_gemm_nt_whole_launches(void);

/* The backward coefficients of the global layer norm in front of the residual layer's heads, WITHOUT a pass over the
 * normalised tensor: with dn = W^T dY (the heads' input gradient) and the heads' weight gradient computed per batch
 * element as G'_b[m][c] = sum_t dY_b[m][t] gamma_c xhat_b[c][t] (ctn_gemm_nt with b_beta = 0),
 *   S1_b = sum_{c,t} gamma_c dn_b[c][t]           = sum_c gamma_c sum_m W[m][c] R_b[m],   R_b[m] = sum_t dY_b[m][t]
 *   S2_b = sum_{c,t} gamma_c dn_b[c][t] xhat[c][t] = sum_{m,c} W[m][c] G'_b[m][c]
 * so c1_b = -rstd_b S1_b / n, c2_b = -rstd_b^2 S2_b / n come from the by-products of the weight-gradient launch
 * (a_rowsum_b, wdot_sums above) and the heads' input-gradient GEMM needs neither the pre-activation tensor nor a
 * statistics epilogue.  The same launch restores the beta term the weight gradient was computed without:
 * dW[m][c] += beta_c sum_b R_b[m] (atomic adds: other streams may be adding to dW).  W / dW: rows m < M1 at
 * W / dW (pitch ld), rows m >= M1 at W2 / dW2 (pitch ld2).  s1acc (batch doubles) and ticket (one 32-bit word) are the
 * call's work area, zeroed by the caller.  Replaces the EPI_NORMBWD statistics of the heads' input-gradient GEMM for
 * gLN; the reference arithmetic is GroupNorm's backward (src/modules/norm.py:18 under autograd). */
int ctn_heads_bwd_coefs(const float* W, int64_t ld, const float* W2, int64_t ld2, int M, int M1, int H,
                        const float* gamma, const float* beta, const float* rowsum_b, const double* wdot_sums,
                        const float* rstd, int batch, double count, float* c1, float* c2, float* dW, int64_t lddw,
                        float* dW2, int64_t lddw2, double* s1acc, unsigned* ticket, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layer-norm statistics (src/modules/norm.py:18 nn.GroupNorm(1, C, eps); :78-90 cumulative form)
 * ------------------------------------------------------------------------------------------- */
/* sums[b][0] += sum prelu(x), sums[b][1] += sum prelu(x)^2 over (C, T)  [gln]   or
 * col[b][0][t] += sum_c prelu(x), col[b][1][t] += sum_c prelu(x)^2     [cln]; alpha NULL = no PReLU */
int ctn_norm_stats(const float* x, int batch, int C, int T, int64_t ld, const float* alpha,
                   double* sums, float* col, int64_t col_ld, void* stream);
/* gLN: mean = S1/count, rstd = 1/sqrt(S2/count - mean^2 + eps), count = C*T */
int ctn_gln_finalize(const double* sums, int batch, double count, float eps, float* mean, float* rstd,
                     void* stream);
/* cLN: prefix sums over t of the column sums, divided by C*(t+1); rstd = 1/(sqrt(var)+eps) */
int ctn_cln_finalize(const float* col, int batch, int C, int T, int64_t col_ld, float eps, float* mean,
                     float* rstd, int64_t stat_ld, void* stream);
/* y = (prelu(x) - mean) * rstd * gamma + beta */
int ctn_norm_apply(const float* x, int batch, int C, int T, int64_t ld, const float* alpha,
                   const float* mean, const float* rstd, int64_t ssb, int64_t sst, const float* gamma,
                   const float* beta, float* y, int64_t ldy, void* stream);
/* reductions of the norm backward: with g = gamma*dn, xhat = (prelu(x)-mean)*rstd:
 *   gln: sums[b][0] += sum g, sums[b][1] += sum g*xhat ; cln: the same per frame into col;
 *   dgamma[c] += sum dn*xhat, dbeta[c] += sum dn */
int ctn_norm_bwd_reduce(const float* dn, int64_t lddn, const float* x, int64_t ld, int batch, int C, int T,
                        const float* alpha, const float* mean, const float* rstd, int64_t ssb, int64_t sst,
                        const float* gamma, double* sums, float* col, int64_t col_ld, float* dgamma,
                        float* dbeta, void* stream);
/* gLN: c1 = -rstd*S1/count, c2 = -rstd^2*S2/count */
int ctn_gln_bwd_coefs(const double* sums, const float* rstd, int batch, double count, float* c1, float* c2,
                      void* stream);
/* cLN: reverse prefix sums of the per-frame mean/variance gradients (derivation in csrc/ctn_norm.hip) */
int ctn_cln_bwd_coefs(const float* col, int64_t col_ld, const float* mean, const float* rstd, int64_t stat_ld,
                      int batch, int C, int T, float eps, float* c1, float* c2, void* stream);
/* dx = (gamma*dn*rstd + c1 + (prelu(x)-mean)*c2) * (x > 0 ? 1 : alpha);  dalpha += sum du*min(x,0) */
int ctn_norm_bwd_apply(const float* dn, int64_t lddn, const float* x, int64_t ld, int batch, int C, int T,
                       const float* alpha, const float* mean, const float* rstd, const float* c1,
                       const float* c2, int64_t ssb, int64_t sst, const float* gamma, float* dx, int64_t lddx,
                       double* dalpha, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused  [PReLU -> layer norm] -> zero-pad -> dilated depthwise Conv1d -> [PReLU statistics]
 * Replaces src/models/tdcn.py:119-132 (PReLU, norm, F.pad of the hidden tensor) and :181-187
 * (depthwise nn.Conv1d(groups=H, dilation=d, bias=True), PReLU, norm statistics); with
 * xflags = 0 and out_alpha = NULL it is the plain depthwise conv of src/modules/conv.py:22,26.
 * The input transform is applied on load, so the normalised tensor never reaches HBM; the output
 * is the pre-activation y2 and the statistics are those of prelu(y2, out_alpha).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t batch, C, T, P;           /* P taps */
    int32_t dilation, pad_left;       /* input index of tap k for output t: t + k*dilation - pad_left */
    int32_t xflags;                   /* CTN_XF_PRELU | CTN_XF_NORM on the input */
    int32_t cln;                      /* 0: esums (gLN) ; 1: ecol (cLN) */
    const float* x; int64_t ldx, xbs;
    const float* alpha_in;
    const float* mean; const float* rstd; int64_t ssb, sst;
    const float* gamma; const float* beta;
    const float* w;                   /* [C][P] */
    const float* bias;                /* [C] or NULL */
    float* y; int64_t ldy, ybs;
    const float* alpha_out;           /* statistics of prelu(y, alpha_out); NULL: no statistics */
    double* esums;                    /* [batch][CTN_STAT_SLOTS][2] += */
    float* ecol; int64_t ecol_ld;     /* [batch][2][ecol_ld] += */
    ctn_finish fin;                   /* esums: finish the reduction in this launch (CTN_FIN_STATS) */
} ctn_dw_fwd_args;

int ctn_dw_fwd(const ctn_dw_fwd_args* args, void* stream);

/* Backward of the above plus both neighbouring norm/PReLU backward pieces:
 *   dy2 = (gamma2*dn2*rstd2 + c1 + (prelu(y2)-mean2)*c2) * (y2 > 0 ? 1 : alpha2)   [if oflags has NORMBWD,
 *          else dy2 = dn2]
 *   dn1[t] = sum_k w[k] * dy2[t - k*dilation + pad_left]            -> dx
 *   dw[c][k] += sum_t dy2[t] * n1[t + k*dilation - pad_left],  dbias[c] += sum_t dy2[t]
 *   dgamma2[c] += sum dn2*xhat2, dbeta2[c] += sum dn2, dalpha2 += sum du2*min(y2,0)
 *   norm-1 backward sums of g1 = gamma1*dn1 and g1*xhat1 (isums / icol), dgamma1, dbeta1.
 * (autograd of src/models/tdcn.py:119-132,181-187) */
typedef struct {
    int32_t batch, C, T, P;
    int32_t dilation, pad_left;
    int32_t xflags;                   /* input-side transform used in forward */
    int32_t oflags;                   /* 0 or CTN_XF_NORMBWD for the output side */
    int32_t cln; int32_t pad0;
    /* output side */
    const float* dn2; int64_t lddn2, dn2bs;
    const float* y2;  int64_t ldy2,  y2bs;
    const float* alpha2;
    const float* mean2; const float* rstd2; const float* c1_2; const float* c2_2; int64_t ssb2, sst2;
    const float* gamma2;
    float* dgamma2; float* dbeta2; double* dalpha2;
    /* input side */
    const float* x; int64_t ldx, xbs;
    const float* alpha1;
    const float* mean1; const float* rstd1; int64_t ssb1, sst1;
    const float* gamma1; const float* beta1;
    const float* w;
    float* dw; float* dbias;          /* [C][P] +=, [C] += (dbias may be NULL) */
    float* dx; int64_t lddx, dxbs;    /* dn1 */
    double* isums; float* icol; int64_t icol_ld;   /* norm-1 backward sums (+=) */
    float* dgamma1; float* dbeta1;
    /* recompute_y2 != 0: y2 is not read (it may be NULL) but recomputed from x, w and `bias` (the forward's bias, NULL =
     * none), which takes one of the four tensors off the kernel's HBM traffic.  Honoured for whole-row gLN launches
     * (oflags NORMBWD, xflags PRELU|NORM, per-sample statistics, T <= 4096, P <= 4); any other launch needs y2. */
    const float* bias;
    int32_t recompute_y2; int32_t pad1;
    ctn_finish fin;                   /* isums: finish the reduction in this launch (CTN_FIN_COEFS) */
} ctn_dw_bwd_args;

int ctn_dw_bwd(const ctn_dw_bwd_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder / decoder (src/models/filterbank.py:205-251)
 * ------------------------------------------------------------------------------------------- */
/* w[b][n][t] = act(sum_l W[n][l] * xp[b][t*S + l]),  xp[tau] = x[b][tau - pad_left] inside [0, Tin) else 0:
 * the F.pad of conv_tasnet.py:149 and nn.Conv1d(1, N, L, stride=S, bias=False) + optional ReLU of
 * filterbank.py:212-230.  x is (batch, Tin) with row stride xbs; L <= 64.  Optionally accumulates the
 * statistics of the separator's entry norm (conv_tasnet.py:370): sums (gLN) or col (cLN), else both NULL. */
int ctn_encoder_fwd(const float* x, int64_t xbs, int Tin, int pad_left, int batch, const float* W, int N, int L,
                    int S, int relu, float* w, int Tf, int64_t ldw, double* sums, float* col, int64_t col_ld,
                    void* stream);

/* xhat[b][s][tau - trim_left] = sum_n sum_{t*S+l = tau} Wd[n][l] * w[b][n][t] * mask[b][s*N+n][t]
 * for 0 <= tau - trim_left < Tout: the mask product of conv_tasnet.py:159-160, nn.ConvTranspose1d(N, 1, L,
 * stride=S, bias=False) of filterbank.py:243-247 and the trim of conv_tasnet.py:169.  mask may be NULL
 * (plain transposed convolution of w, n_src = 1).  L <= 32, stride such that (L-1)/S < 32; n_src <= 4 (<= 3 when L > 16). */
int ctn_maskdecode_fwd(const float* w, int64_t ldw, const float* mask, int64_t ldm, int batch, int n_src,
                       int N, int Tf, const float* Wd, int L, int S, float* out, int64_t out_ss,
                       int64_t out_bs, int Tout, int trim_left, void* stream);

/* Backward of ctn_maskdecode_fwd with a sigmoid mask:
 *   dwh = sum_l Wd[n][l]*dout[b][s][t*S+l-trim_left];  dz = dwh*w*m*(1-m) (sigmoid_bwd=1) or dwh*w;
 *   dw_dec[b][n][t] = sum_s dwh*m;   dWd[n][l] += sum w*m*dout.   L <= 32, S <= 16, n_src <= 4 (<= 3 when L > 16).
 * dz may be NULL when mask is NULL. */
int ctn_maskdecode_bwd(const float* dout, int64_t dout_ss, int64_t dout_bs, int Tout, int trim_left,
                       const float* w, int64_t ldw, const float* mask, int64_t ldm, int batch, int n_src,
                       int N, int Tf, const float* Wd, int L, int S, int sigmoid_bwd, float* dz, int64_t lddz,
                       float* dw_dec, int64_t lddw, float* dWd, void* stream);

/* Encoder backward: dwt = [relu ? (w > 0) : 1] * (dw_a + norm_bwd(dn, w)) ; dW[n][l] += sum_t dwt*xp[t*S+l];
 * dgamma[n] += sum dn*xhat ; dbeta[n] += sum dn.  dw_a may be NULL; dn and the norm arguments may be NULL
 * (no entry norm).  dwt_out (batch, N, lddwt), if not NULL, receives dwt (the gradient w.r.t. the
 * encoder's pre-activation output).  L <= 32, S <= 16. */
int ctn_encoder_bwd(const float* x, int64_t xbs, int Tin, int pad_left, int batch, int N, int L, int S, int relu,
                    const float* w, int Tf, int64_t ldw, const float* dw_a, int64_t lddwa, const float* dn,
                    int64_t lddn, const float* mean, const float* rstd, const float* c1, const float* c2,
                    int64_t ssb, int64_t sst, const float* gamma, float* dW, float* dgamma, float* dbeta,
                    float* dwt_out, int64_t lddwt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depthwise Conv1d with arbitrary stride, dilation and zero padding: the depthwise half of the generic
 * DepthwiseSeparableConv1d (src/modules/conv.py:13-29; nn.Conv1d(C, C, K, stride, padding, dilation, groups=C), default
 * stride = kernel_size).  The TCN's stride-1 'same' layers use ctn_dw_fwd / ctn_dw_bwd instead.
 * ------------------------------------------------------------------------------------------- */
/* y[b][c][t] = bias[c] + sum_k w[c*K + k] * x[b][c][t*stride + k*dilation - pad_left]  (zero outside 0..Tin-1),
 * t < Tout; rows of x / y at b*bs + c*ld.  bias may be null. */
int ctn_dwconv_generic_fwd(const float* x, int64_t xbs, int64_t ldx, int batch, int C, int Tin, const float* w,
                           const float* bias, int K, int stride, int dilation, int pad_left, float* y, int64_t ybs,
                           int64_t ldy, int Tout, void* stream);
/* dx (written, may be null), dW[C*K] and dbias[C] (accumulated into: zero them first; either may be null). */
int ctn_dwconv_generic_bwd(const float* x, int64_t xbs, int64_t ldx, int batch, int C, int Tin, const float* w, int K,
                           int stride, int dilation, int pad_left, const float* dy, int64_t dybs, int64_t lddy, int Tout,
                           float* dx, int64_t dxbs, int64_t lddx, float* dW, float* dbias, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Utterance-level PIT SI-SDR (src/criterion/pit.py:9-44 over src/criterion/sdr.py:122-139,198-227)
 * ------------------------------------------------------------------------------------------- */
/* work: double[batch * (2*S*S + S)] zeroed by the caller.  pair[b][i][j] = SI-SDR(est_i, tgt_j) in dB.
 * loss_b[b] = min over permutations p (itertools order, first minimum wins) of -mean_i pair[i][p(i)];
 * perm[b][i] = p(i) (int64); coef[b][i][2] = (a, c) with d loss_b / d est_i = a*tgt_{p(i)} + c*est_i.
 * fixed_perm != 0 skips the search and uses the identity assignment (plain NegSISDR, sdr.py:198-227).
 * S <= 4. */
int ctn_pit_sisdr_fwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs,
                      float eps, int fixed_perm, double* work, float* pair, float* loss_b, int64_t* perm,
                      float* coef, void* stream);
/* ctn_pit_sisdr_fwd without the `work` argument.  No counterpart in the reference: it is for callers that cannot
 * allocate device memory or bring a workspace -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  The sums live in a device array built into the library (CTN_PIT_NOALLOC_MAX_WORK doubles = 512 KiB,
 * held by every GPU the library is used on until the process ends); the call clears the part it uses on `stream` and
 * then enqueues the same three kernels.  It calls no allocator.  batch * (2*S*S + S) beyond the array is rejected with
 * CTN_ERR_ARG.  All calls on one GPU share the array: calls that overlap on one GPU (different streams, or threads)
 * race; serialise them or use ctn_pit_sisdr_fwd.  The first call on a device resolves the array's device address
 * (hipGetSymbolAddress), so make that one outside graph capture. */
#define CTN_PIT_NOALLOC_MAX_WORK 65536
int ctn_pit_sisdr_fwd_noalloc(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs,
                              float eps, int fixed_perm, float* pair, float* loss_b, int64_t* perm, float* coef,
                              void* stream);
/* dest[b][i][t] = gscale[b*gs_bstride + i*gs_istride] * (a*tgt[b][p(i)][t] + c*est[b][i][t]) */
int ctn_pit_sisdr_bwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs,
                      const int64_t* perm, const float* coef, const float* gscale, int gs_bstride, int gs_istride,
                      float* dest, int64_t dss, int64_t dbs, void* stream);

/* The whole S x S matrix with its derivative, for the plain SDR (src/criterion/sdr.py:6-22, :24-114), for permutation
 * searches over it and for Sinkhorn PIT (src/criterion/pit.py:163-193), which weights every pair.
 * kind 0: pair[b][i][j] = SI-SDR(est_i, tgt_j); kind 1: pair[b][i][j] = 10 log10((|tgt_j|^2 + eps) / (|tgt_j - est_i|^2
 * + eps)).  paircoef[b][i][j][2] = (ct, ce) with d pair[b][i][j] / d est_i = ct*tgt_j + ce*est_i.
 * work: double[batch * (2*S*S + S)] zeroed by the caller.  S <= 4. */
int ctn_pair_sdr_fwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs, float eps,
                     int kind, double* work, float* pair, float* paircoef, void* stream);
/* dest[b][i][t] = sum_j A[b][i][j]*tgt[b][j][t] + C[b][i]*est[b][i][t]   (A = upstream gradient x ct, C = sum_j of
 * upstream gradient x ce) */
int ctn_pair_sdr_bwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs,
                     const float* A, const float* C, float* dest, int64_t dss, int64_t dbs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step on flat fp32 buffers (egs/wsj0-mix/common/src/driver.py:152-155:
 * clip_grad_norm_(max_norm) then torch.optim.Adam.step)
 * ------------------------------------------------------------------------------------------- */
/* acc[0] += sum g^2 */
int ctn_sumsq(const float* g, int64_t n, double* acc, void* stream);
/* clip = max_norm > 0 ? min(1, max_norm/(sqrt(sumsq[0]) + 1e-6)) : 1;  g' = clip * grad_scale * g;
 * Adam: m = b1*m + (1-b1)*g'; v = b2*v + (1-b2)*g'^2; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps) */
int ctn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq,
                  float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps, int step,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTN_H */
