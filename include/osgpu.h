/*
 * osgpu.h -- C ABI of the MI355X (gfx950) operator backend that takes the seat of OnnxStream's `class XnnPack`
 * (reference src/onnxstream.cpp:657-2150) and of the inline host kernels `Model::run` launches through
 * `XnnPack::parallelize` (call sites :4130 Erf/Sqrt/Sin/Cos, :4293 Concat, :5046 InstanceNormalization,
 * :5384 ReduceMean, :5595 Pow, :6110 Split, :6309 Resize, :6489 Gather, :6670 Slice).
 *
 * Conventions
 *  - plain C: opaque context, raw DEVICE pointers (void*), sizes; no C++/torch types cross this boundary.
 *  - every compute entry point is ASYNCHRONOUS on the context's compute stream (like the reference's
 *    CublasOps precedent, onnxstream.cpp:308-352); `osg_download` / `osg_sync` are the sync points
 *    (== CublasOps::ensure_is_ready, onnxstream.cpp:200-230).
 *  - return value: 0 = success, non-zero = failure; the message is available from osg_last_error()
 *    (the host Model turns it into the std::runtime_error/invalid_argument the reference throws).
 *  - element types: OSG_F16 (IEEE half, the reference's uint16_t "fp16 bits"), OSG_F32, OSG_U8, OSG_I64.
 *  - arithmetic contract (matches the XNNPACK CPU path the reference uses): f16 operands, f32 accumulation /
 *    f32 intermediate math, ONE round-to-nearest-even to f16 on store of every entry point's output.
 *  - tensors are dense row-major; 4-D activations around convolutions are NHWC exactly like the reference
 *    (XnnPack::convolution returns [1,Ho,Wo,Cout], onnxstream.cpp:1292-1534).
 *  - `batch`/N arguments generalise the reference's per-op batch loop (onnxstream.cpp:3847): N independent samples
 *    that share one weight fetch are executed as one launch.
 */
#ifndef OSGPU_H
#define OSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct osg_ctx osg_ctx;
typedef struct osg_graph osg_graph;

typedef enum { OSG_U8 = 1, OSG_F16 = 2, OSG_F32 = 3, OSG_I64 = 4 } osg_dtype; /* == onnxstream::TensorDataType */

typedef enum {
  OSG_ACT_NONE = 0,
  OSG_ACT_SILU = 1,
  OSG_ACT_SIGMOID = 2,
  /* osg_gemm only: C[M,N/2] = v * gelu_erf(g) with v/g the "value"/"gate" halves of the projection (+bias); the [N,K] weight and
   * the bias must be pair-interleaved in blocks of 16 rows: rows 32k..32k+15 = value columns 16k.., rows 32k+16..32k+31 = the
   * matching gate columns (the host Model builds this layout once when the weight becomes resident). */
  OSG_ACT_GEGLU = 3
} osg_act;

typedef enum {
  OSG_UN_SIGMOID = 0, /* XnnPack::sigmoid            onnxstream.cpp:1217 */
  OSG_UN_ERF = 1,     /* Model::run Erf  (std::erf)  onnxstream.cpp:4001-4139 */
  OSG_UN_SQRT = 2,    /* Model::run Sqrt             onnxstream.cpp:4001-4139 */
  OSG_UN_SIN = 3,     /* Model::run Sin              onnxstream.cpp:4001-4139 */
  OSG_UN_COS = 4,     /* Model::run Cos              onnxstream.cpp:4001-4139 */
  OSG_UN_NEG = 5,     /* Model::run Neg              onnxstream.cpp:7475 */
  OSG_UN_POW = 6,     /* Model::run Pow (scalar exponent in `param`) onnxstream.cpp:5478-5604 */
  OSG_UN_SILU = 7,    /* fused Sigmoid+Mul (x*sigmoid(x)) */
  OSG_UN_GELU_ERF = 8 /* fused Div,Erf,Add,Mul,Mul (0.5*x*(1+erf(x/sqrt2))) */
} osg_unary_kind;

typedef enum {
  OSG_BIN_ADD = 0, /* XnnPack::add       onnxstream.cpp:1666 */
  OSG_BIN_SUB = 1, /* XnnPack::subtract  onnxstream.cpp:1811 */
  OSG_BIN_MUL = 2, /* XnnPack::multiply  onnxstream.cpp:846  */
  OSG_BIN_DIV = 3  /* XnnPack::divide    onnxstream.cpp:1881 */
} osg_binary_kind;

/* ---- lifetime / device ---------------------------------------------------------------------------- */
int osg_device_count(void);
int osg_init(int device, osg_ctx** out);  /* replaces XnnPack::XnnPack(threads) onnxstream.cpp:678 */
void osg_destroy(osg_ctx* ctx);           /* replaces XnnPack::~XnnPack          onnxstream.cpp:689 */
const char* osg_last_error(const osg_ctx* ctx);
const char* osg_device_name(const osg_ctx* ctx);
void* osg_stream(const osg_ctx* ctx);     /* the compute hipStream_t (for callers that record their own events) */
/* on != 0: the first eager launch of each GEMM / convolution shape TIMES the legal tile / split-K configurations on the caller's operands
 * and later launches (graph captures included) reuse the fastest; the choice is shared by every context of the process on that device.
 * Takes the seat of XNNPACK's per-operator microkernel selection at xnn_create_* time (onnxstream.cpp:1104-1182).  Default off. */
int osg_set_autotune(osg_ctx* ctx, int on);
/* Shapes an autotuning context looked up in the measured-choice table (OSG_TUNE_CACHE) and did not find, since the process started: with OSG_TUNE_FROZEN=1 such a
 * shape runs the cost model's first candidate untimed -- a job whose ranks must time the SAME plan (bench.py --gpus N) wants this to be 0. */
int osg_tune_misses(void);

/* ---- memory / transfers (CublasOps buffer pool + cudaMemcpyAsync precedent, onnxstream.cpp:141-230,325,347) --- */
int osg_malloc(osg_ctx* ctx, size_t bytes, void** dptr);
int osg_free(osg_ctx* ctx, void* dptr);
int osg_upload(osg_ctx* ctx, void* dst, const void* host_src, size_t bytes);         /* pinned staging + async H2D on the COPY stream; compute stream waits on it */
int osg_upload_sync(osg_ctx* ctx, void* dst, const void* host_src, size_t bytes);    /* plain blocking H2D */
/* streamed-weights mode: page-lock a weights provider's own host buffer once, then DMA from it with no staging copy
 * (async on the COPY stream; the compute stream waits on the copy's event).  The caller keeps the memory alive and registered. */
int osg_host_register(osg_ctx* ctx, void* host_ptr, size_t bytes);
int osg_host_unregister(osg_ctx* ctx, void* host_ptr);
int osg_upload_pinned(osg_ctx* ctx, void* dst, const void* pinned_host_src, size_t bytes);
/* the streamed pass's form of it: enqueue only, alternating between two H2D queues (OSG_COPY_STREAMS=1: one); osg_copy_fence makes the compute
 * stream wait for everything enqueued so far -- called once per step, after the uploads of all the weights the step reads. */
int osg_upload_pinned_async(osg_ctx* ctx, void* dst, const void* pinned_host_src, size_t bytes);
int osg_copy_fence(osg_ctx* ctx);
int osg_download(osg_ctx* ctx, void* host_dst, const void* src, size_t bytes);       /* D2H + wait == ensure_is_ready */
/* VRAM-budgeted weight streaming (CudaOptions::m_vram_to_use, onnxstream.cpp:396-398): device buffers of the streaming ring are recycled, so
 * an upload into one must not start before the LAUNCHES that read its previous occupant have finished.  osg_marker_record(slot) marks the
 * current position of the COMPUTE stream; osg_copy_wait_marker(slot) makes the COPY stream (every later osg_upload*) wait for it.
 * slot in [0, 256). */
int osg_marker_record(osg_ctx* ctx, int slot);
int osg_copy_wait_marker(osg_ctx* ctx, int slot);
int osg_copy(osg_ctx* ctx, void* dst, const void* src, size_t bytes);                /* async D2D on compute stream */
int osg_memset(osg_ctx* ctx, void* dst, int value, size_t bytes);
int osg_sync(osg_ctx* ctx);

/* ---- stream capture: a whole Model::run pass replayed as one hipGraph ------------------------------------ */
int osg_graph_begin(osg_ctx* ctx);
int osg_graph_end(osg_ctx* ctx, osg_graph** out);
int osg_graph_launch(osg_ctx* ctx, osg_graph* g);
void osg_graph_destroy(osg_graph* g);

/* ---- timing on the compute stream (HIP events) ------------------------------------------------------------ */
int osg_timer_start(osg_ctx* ctx);
int osg_timer_stop(osg_ctx* ctx, float* ms); /* waits for the stop event */
/* a sequence of timestamps on the compute stream WITHOUT host synchronisation in between: osg_timer_mark(i) records event i (i < 4096) behind
 * everything enqueued so far; osg_timer_between(a, b) waits for event b and returns the device time from a to b.  Lets a profiler time every
 * launch of a pass while the queue stays full (per-launch figures comparable with rocprofv3's kernel durations, no idle-launch latency). */
int osg_timer_mark(osg_ctx* ctx, int index);
int osg_timer_between(osg_ctx* ctx, int a, int b, float* ms);

/* ---- tracing: rocTX ranges around the launches of one graph op (visible in rocprofv3 --marker-trace timelines); no-ops when libroctx64 is absent.
 * The host Model opens a range per step when m_ops_printf / m_ops_times_printf is set or OSG_ROCTX=1 (the reference's tracing is the two printf
 * flags, onnxstream.cpp:3759-3762, :8199-8214). */
void osg_range_push(const char* name);
void osg_range_pop(void);

/* ---- dense contractions ------------------------------------------------------------------------------------ */
/* Convolution, NHWC in / OHWI weights / NHWC out, group 1, dilation 1, f32 accumulate.
 * Replaces XnnPack::convolution<T,U> (onnxstream.cpp:1292-1534).  `bias` (f16 or f32 per bias_dtype, may be NULL),
 * optional fused `residual` (same shape as y, added in f32 before the single rounding) and activation. */
int osg_conv2d_nhwc(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w_ohwi, const void* bias, osg_dtype bias_dtype,
                    const void* residual, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h,
                    int stride_w, int pad_top, int pad_left, int pad_bottom, int pad_right, osg_act act);

/* Same, plus a per-IMAGE channel bias image_bias[n*image_bias_ld + c] (f16) added in f32 before the single rounding: the fused form
 * of the resnet block's `Conv -> Add(Unsqueeze(Unsqueeze(time_emb_proj)))` (reference: XnnPack::add with a [1,C,1,1] operand, :1666). */
int osg_conv2d_nhwc_rb(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w_ohwi, const void* bias, osg_dtype bias_dtype,
                       const void* image_bias, long image_bias_ld, const void* residual, void* y, int N, int H, int W, int Cin, int Cout,
                       int KH, int KW, int stride_h, int stride_w, int pad_top, int pad_left, int pad_bottom, int pad_right, osg_act act);

/* Same, with output VIEWS: the reference materialises every Concat with a copy (onnxstream.cpp:4140-4299); here the convolution that produces a
 * skip tensor can store its result straight into the column slice of the concatenated NHWC buffer the up-block will read.
 *   y_ld   row pitch of y in ELEMENTS (0 = Cout, dense): pixel p's channels start at y + p * y_ld;
 *   y2     optional second destination (NULL = none) receiving the same f16 values, rows y2_ld elements apart -- the dense tensor for the layers
 *          that read it as it stands plus the slice of the concatenation, in one launch.
 * Pitches must be multiples of 4 elements (8-byte stores) when Cout is; residual / image_bias / activation as in osg_conv2d_nhwc_rb. */
int osg_conv2d_nhwc_v(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w_ohwi, const void* bias, osg_dtype bias_dtype,
                      const void* image_bias, long image_bias_ld, const void* residual, void* y, long y_ld, void* y2, long y2_ld, int N, int H, int W,
                      int Cin, int Cout, int KH, int KW, int stride_h, int stride_w, int pad_top, int pad_left, int pad_bottom, int pad_right,
                      osg_act act);

/* C[b] = act(A[b] (MxK, row-major, lda) * B[b] + bias + residual).  B is [K,N] row-major (b_is_nk=0, the layout
 * XNN_FLAG_TRANSPOSE_WEIGHTS gives the reference, onnxstream.cpp:977,1136) or pre-transposed [N,K] (b_is_nk=1, how
 * resident weights are kept on the device).  stride_* are element strides between batch items (0 = broadcast).
 * Replaces XnnPack::matrix_multiply / matrix_multiply_dynamic (onnxstream.cpp:929-1215) and the n-loop of
 * Model::run MatMul (:5798).  bias: [N] (row broadcast) ; residual: [batch,M,N] like C. */
int osg_gemm(osg_ctx* ctx, osg_dtype dtype, const void* A, const void* B, int b_is_nk, const void* bias, osg_dtype bias_dtype,
             const void* residual, void* C, int M, int N, int K, int batch, long stride_a, long stride_b, long stride_c,
             osg_act act);

/* W8A16: the same contractions with the weight operand resident as uint8 codes + (scale, zero_point).  The reference dequantises a uint8
 * weight when it LOADS it, w = f16((float)((int)q - zero_point) * scale) (get_tensor_data :2887-2891 -> Model::dequantize :3353), and runs the
 * f16 contraction; here the codes stay uint8 in HBM, stream through L2 and the LDS ring as codes (half the bytes of the weight operand on
 * every hop) and become halves between the LDS tile and the MFMA: the integer q - zero_point EXACTLY, the scale applied once to the f32
 * accumulator -- sum_k a[m][k] (q[n][k] - zp[n]) * scale[n].  The result differs from the reference's by the rounding the reference applies
 * to each dequantised weight (2^-12 relative per term) and is the closer of the two to the f32 contraction.  Same tuned kernels, tiles,
 * split-K, GEGLU epilogue (pair-interleaved codes), output views and statistics sinks as the f16 entry points.
 * Bq_nk is [N,K] (K contiguous), wq_ohwi is [Cout,KH,KW,Cin]; K (Cin) must be a multiple of 64, zero_point in 0 .. 255.
 * _v: per-output-column quantisation parameters as device vectors [N] of float (both or neither; N % 4 == 0; they override the scalars) --
 * the merged projections concatenate weights quantised one by one --, and for the convolution the output views of osg_conv2d_nhwc_v. */
int osg_gemm_w8(osg_ctx* ctx, const void* A, const void* Bq_nk, float w_scale, int w_zero_point, const void* bias, osg_dtype bias_dtype,
                const void* residual, void* C, int M, int N, int K, osg_act act);
int osg_conv2d_nhwc_w8(osg_ctx* ctx, const void* x, const void* wq_ohwi, float w_scale, int w_zero_point, const void* bias,
                       osg_dtype bias_dtype, const void* image_bias, long image_bias_ld, const void* residual, void* y, int N, int H, int W,
                       int Cin, int Cout, int KH, int KW, int stride_h, int stride_w, int pad_top, int pad_left, int pad_bottom,
                       int pad_right, osg_act act);
int osg_gemm_w8_v(osg_ctx* ctx, const void* A, const void* Bq_nk, float w_scale, int w_zero_point, const float* w_scale_vec,
                  const float* w_zero_point_vec, const void* bias, osg_dtype bias_dtype, const void* residual, void* C, int M, int N, int K,
                  osg_act act);
int osg_conv2d_nhwc_w8_v(osg_ctx* ctx, const void* x, const void* wq_ohwi, float w_scale, int w_zero_point, const float* w_scale_vec,
                         const float* w_zero_point_vec, const void* bias, osg_dtype bias_dtype, const void* image_bias, long image_bias_ld,
                         const void* residual, void* y, long y_ld, void* y2, long y2_ld, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride_h, int stride_w, int pad_top, int pad_left, int pad_bottom, int pad_right, osg_act act);

/* Re-layout a [K,N] row-major matrix into [N,K] (done once per resident weight). */
int osg_transpose_kn_to_nk(osg_ctx* ctx, osg_dtype dtype, const void* src_kn, void* dst_nk, int K, int N);
/* LayerNorm folded into the GEMM that consumes it (the decomposed LayerNorm chain, onnxstream.cpp:5237-5604, followed by MatMul :5669):
 *   y[m][n] = act( rstd_m * ( sum_k x[m][k] * W'[n][k]  -  mean_m * c1[n] )  +  c2[n]  (+ residual[m][n]) )
 * with mean_m / rstd_m = LayerNorm statistics of row m of x over K (eps inside the sqrt), W'[n][k] = f16(gamma[k] * W[n][k]) prepared by
 * the caller, c1[n] = sum_k W'[n][k], c2[n] = sum_k beta[k] * W[n][k] + bias[n] (fp32 device vectors).  Algebraically
 * LayerNorm(x) . W^T + bias without materialising (or f16-rounding) the normalised activation; the row sums and sums of squares are
 * accumulated in fp32 beside the MFMAs from the very A fragments they consume (no second pass over x), the variance is formed in f64.
 * K % 64 == 0, N % 4 == 0; act may be OSG_ACT_GEGLU (pair-interleaved W', c1, c2).
 * rowstats (may be NULL): partial row statistics of x emitted by the GEMM that produced it (osg_gemm_rowstats) -- [M][K/32][2] floats,
 * (sum, sum of squares) of each 32-column slot; with them this GEMM does no statistics work of its own (K <= 1280). */
int osg_gemm_ln(osg_ctx* ctx, const void* x, const void* w_nk_folded, const float* c1, const float* c2, float eps, const float* rowstats,
                const void* residual, void* y, int M, int N, int K, osg_act act);
/* osg_gemm ([N,K] weight, batch 1) whose epilogue also emits rowstats[M][N/32][2] = (sum, sum of squares) of the f16-ROUNDED outputs over
 * every 32-column slot of every row: the hand-over to an osg_gemm_ln that normalises this output.  N % 32 == 0, K % 64 == 0. */
int osg_gemm_rowstats(osg_ctx* ctx, const void* A, const void* B_nk, const void* bias, osg_dtype bias_dtype, const void* residual, void* C,
                      int M, int N, int K, osg_act act, float* rowstats);

/* Fused attention == the reference's AttentionFusedOps pseudo-op (onnxstream.cpp:6696-6929):
 * for each of `heads` items: O = softmax(scale * Q K^T) V, with Q:[heads,Tq,D], K given TRANSPOSED as the reference
 * receives it (k_is_dt=1: [heads,D,Tkv], :6792,6814) or natural [heads,Tkv,D] (k_is_dt=0), V:[heads,Tkv,D], O:[heads,Tq,D].
 * `scale` multiplies the f16-rounded scores (the reference's separate Mul); scores and probabilities are kept in
 * f32 on chip (flash-style), i.e. FEWER roundings than the sliced reference path.  The strided variant reads Q/K/V
 * directly out of [T, heads*D] projection outputs (token stride ld*, head stride D) so the head split/merge
 * Reshape/Transpose ops vanish. */
int osg_attention(osg_ctx* ctx, osg_dtype dtype, const void* q, const void* k, const void* v, void* o, int heads, int Tq, int Tkv,
                  int D, float scale, int k_is_dt);
int osg_attention_strided(osg_ctx* ctx, osg_dtype dtype, const void* q, long q_tok, long q_head, long q_batch, const void* k,
                          long k_tok, long k_head, long k_batch, const void* v, long v_tok, long v_head, long v_batch, void* o,
                          long o_tok, long o_head, long o_batch, int batch, int heads, int Tq, int Tkv, int D, float scale);
/* The ROW-LOCAL tail of a BasicTransformerBlock as ONE launch (osg_tchain.hip): everything behind the self-attention depends on its own token
 * row only (cross-attention's K / V are the shared text context), so one workgroup carries 64 rows through
 *   x1 = a1 . Wo1^T + bo1 + x0                  MatMul + Add + Add                      (onnxstream.cpp:5669-5861, :3906-4000)
 *   q  = LayerNorm(x1; g2, be2) . Wq2^T         ReduceMean .. Add chain + MatMul        (:5237-5604)
 *   a2 = softmax(scale q k^T) v   per head      AttentionFusedOps                       (:6696-6929)
 *   x2 = a2 . Wo2^T + bo2 + x1
 *   x3 = GEGLU(LayerNorm(x2; g3, be3) . W1^T + b1) . W2^T + b2 + x2     (Slice, Slice, Div, Erf :4001-4139, Add, Mul, Mul, Mul)
 *   y  = x3 . Wpo^T + bpo + xin                 the 1x1 proj_out Conv (:4494-4707) + the spatial residual; only when wpo != NULL, else y = x3
 * with the row block resident in LDS and the GEGLU activation never formed.  All tensors f16; the seven weights in the kn8 layout of osg_tblock_pack_weight
 * ([K/8][N][8]; w1 from [8C][C] = value rows then gate rows, w2 from [C][4C]); a1 / x0 / xin dense [M][C]; out rows ldo apart (0 = C), out2 (may be NULL) a second copy rows ldo2 apart (the Concat
 * slot of a skip connection); kp / vtp from osg_tblock_kv_pack.  M rows = images x rows_per_img, a row block (64 or 32 rows) lies inside one image.
 * rows_per_block: 0 = the library picks (32-row blocks while 64-row blocks would leave CUs without one), 32 / 64 = forced (tests, probes).
 * dbg[0..6] (may be NULL): dense [M][C] dumps of x1, LN(x1), q, a2, x2, LN(x2), x3 (x3 only with wpo) -- the kernel tests read them; dbg[7] (may be NULL):
 * [M/32][32] int64 wall-clock stamps (100 MHz) of every row block's stages -- tools/tblock_tail_probe.py reads them. */
typedef struct {
  const void *a1, *x0;
  const void *wo1, *bo1;
  const void *g2, *be2;
  float eps2;
  const void *wq2, *bq2;
  const void *kp, *vtp;
  float scale;
  int Tk;
  const void *wo2, *bo2;
  const void *g3, *be3;
  float eps3;
  const void *w1, *b1, *w2, *b2;
  const void *wpo, *bpo, *xin;
  void *out, *out2;
  long ldo, ldo2;
  int M, rows_per_img, C, heads;
  void* dbg[8];
  int rows_per_block;
} osg_tblock_tail_args;
int osg_tblock_tail_supported(int M, int rows_per_img, int C, int heads, int Tk); /* 1 = osg_tblock_tail takes the shape */
int osg_tblock_tail(osg_ctx* ctx, const osg_tblock_tail_args* a);
/* A resident [N][K] weight (k contiguous: a MatMul's [K,N] after osg_transpose_kn_to_nk, a 1x1 convolution's OHWI) -> the layout osg_tblock_tail streams:
 * [K/8][N][8], i.e. for every 8-deep k chunk the N rows side by side -- an MFMA fragment request (lane = row, lane group = k chunk) is then four runs of
 * 256 contiguous bytes.  Done once per weight, when it becomes resident.  K % 8 == 0. */
int osg_tblock_pack_weight(osg_ctx* ctx, const void* w_nk, int N, int K, void* w_kn8);
/* K / V of cross-attentions re-packed for osg_tblock_tail, several blocks per launch.  All of them are column ranges of ONE matrix `base` ([imgs * Tk] rows
 * ld apart -- the merged K|V projection of the text context): job j = jobs_dev[4 j .. 4 j + 3] = {column of K, column of V, head dim D, element offset of
 * its packs inside dst}; head h sits h D columns further in.  At dst + offset: kp [img][head][80][DP], then vtp [img][head][DP][80] (V transposed),
 * DP = D rounded up to 16, zero padding.  ld, the columns and D are multiples of 8 (16-byte chunks).  osg_tblock_kv_pack_elems = f16 elements of EACH of the two packs of a job. */
size_t osg_tblock_kv_pack_elems(int imgs, int heads, int D);
int osg_tblock_kv_pack_jobs(osg_ctx* ctx, const void* base, long ld, int imgs, int Tk, int heads, int njobs, const int* jobs_dev, void* dst);
/* ScaledDotProductAttention == the reference's pseudo-op of that name (formed at run time from Transpose/MatMul/Div/Add/Softmax/MatMul or
 * Transpose/Mul/Mul/MatMul/Add/Softmax/MatMul when m_use_scaled_dp_attn_op, onnxstream.cpp:3635-3755; executed at :7767-7882 through
 * XnnPack::scaled_dot_product_attention :2054-2150): out = softmax(scale * q k^T + mask) v per batch and head.  Dense q [B][Hq][Tq][D],
 * k / v [B][Hkv][Tkv][D] (K in its natural layout: the op takes the Transpose's INPUT), mask [Tq][Tkv] additive, shared by all batches and
 * heads, or NULL; out [B][Hq][Tq][D].  Hq % Hkv == 0 (query head h reads key/value head h / (Hq/Hkv)).  `scale` is the f16-rounded
 * 1/s (Div form) or s*s2 (Mul/Mul form) the reference computes at :7840-7862.  Same flash-style kernel as osg_attention (f32 scores). */
/* The two elementwise chains of the LLM graphs (reference: run op by op through XnnPack::multiply / add / ... and the inline kernels), fused:
 * RMSNorm = Pow :5478 -> ReduceMean :5237 -> Add -> Sqrt :4001 -> Div -> Mul -> Mul under m_requires_upcast (fp32 from the f16 input to ONE rounding);
 * rotary embedding = Slice :6499 x2 -> Neg :7475 -> Concat :4140 -> Mul x2 -> Add with the chain's own f16 roundings (bit-identical). */
int osg_rms_norm(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w, void* y, long rows, int C, float eps);
int osg_rope(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* cos_t, const void* sin_t, void* y, long bh, long T, int d);
int osg_sdpa(osg_ctx* ctx, osg_dtype dtype, const void* q, const void* k, const void* v, const void* mask, void* o, int batch, int q_heads,
             int kv_heads, int Tq, int Tkv, int D, float scale);

/* ---- normalisation / reductions ---------------------------------------------------------------------------- */
/* InstanceNormalization on [rows, L] (reference input [1,G,L], onnxstream.cpp:4788-5055): per row mean/var in f32,
 * y = scale[row % n_scale]*(x-mean)/sqrt(var+eps)+bias[row % n_scale]; scale/bias are f32 (forced f32 at :4802). */
int osg_instance_norm(osg_ctx* ctx, osg_dtype dtype, const void* x, const float* scale, const float* bias, void* y, int rows,
                      long L, int n_scale, float eps);
/* Fused GroupNorm on NHWC [N,HW,C] = Reshape->InstanceNorm->Reshape->Mul(gamma[C])->Add(beta[C]) (+SiLU). */
int osg_group_norm_nhwc(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* gamma, const void* beta, void* y, int N, long HW,
                        int C, int groups, float eps, osg_act act);
/* GroupNorm statistics from the PRODUCER: osg_set_stat_sinks arms the NEXT osg_conv2d_nhwc_v -- its launch adds, per (image, group), the sum and the sum
 * of squares of the f16 values it stores to table0 (for its output y) / table1 (for its second destination y2): tables [8][N][groups][2] of int64 fixed-point
 * sums (integer atomics: the same bits whatever order the workgroups arrive in; eight copies, a workgroup adds to the copy of the XCD it runs on with an
 * atomic executed in that XCD's L2, the reader sums them), zeroed by the caller before the first producer of a pass.  cpg = channels
 * per group, ch_off = channel offset of this convolution's output inside the tensor the GroupNorm normalises (a Concat slot), rows_per_image = Ho * Wo
 * (a multiple of 128).  The kernel's epilogue serves the sinks where it can (one k-slice, 4-aligned shapes), a small launch of its own otherwise.
 * osg_group_norm_stats_nhwc then normalises from the table in ONE streaming launch (the reference's GroupNorm = InstanceNormalization over [1,G,L]
 * + affine, onnxstream.cpp:4788-5055, as osg_group_norm_nhwc).  f16 only. */
int osg_set_stat_sinks(osg_ctx* ctx, void* table0, int groups0, int cpg0, int ch_off0, void* table1, int groups1, int cpg1, int ch_off1, int rows_per_image);
int osg_group_norm_stats_nhwc(osg_ctx* ctx, const void* x, const void* gamma, const void* beta, void* y, int N, long HW, int C, int G, float eps, osg_act act,
                              const void* stat_table);
/* Fused LayerNorm over the last axis == ReduceMean,Sub,Pow,ReduceMean,Add,Sqrt,Div,Mul,Add (onnxstream.cpp:5237-5604). */
int osg_layer_norm(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* gamma, const void* beta, void* y, long rows, int C,
                   float eps);
/* ReduceMean over the last axis, keepdims (onnxstream.cpp:5237-5393). */
int osg_reduce_mean_last(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, long rows, long C);
/* Softmax over the last axis (XnnPack::softmax, onnxstream.cpp:1958). */
int osg_softmax_last(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, long rows, long C);

/* ---- elementwise -------------------------------------------------------------------------------------------- */
int osg_unary(osg_ctx* ctx, osg_dtype dtype, osg_unary_kind kind, const void* x, void* y, long n, float param);
/* NumPy-style broadcasting binary op, shapes right-aligned to `rank` (<=6) dims (onnxstream.cpp:855-876). */
int osg_binary(osg_ctx* ctx, osg_dtype dtype, osg_binary_kind kind, const void* a, const long* a_shape, const void* b,
               const long* b_shape, void* y, int rank);
/* GEGLU: x:[rows,2C] -> y:[rows,C] = x[:, :C] * gelu_erf(x[:, C:]) (Slice,Slice,Div,Erf,Add,Mul,Mul,Mul). */
int osg_geglu(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, long rows, long C);

/* ---- denoising-loop glue on the device -------------------------------------------------------- */
/* CFGDenoiser input side (src/sd.cpp:1427-1470): sample[2p] = sample[2p+1] = x[p] * c_in for p < prompts (L floats each, fp32);
 * timestep[0 .. 2*prompts*t_per_sample) = t.  x, sample, timestep are DEVICE fp32 buffers (the plan's input staging). */
int osg_sampler_prepare(osg_ctx* ctx, const float* x, float* sample, float* timestep, int prompts, long L, float c_in, float t,
                        long t_per_sample);
/* One Euler-Ancestral step with the CFG combine (src/sd.cpp:1545-1556, src/samplers.h:1430-1449), in place on x:[prompts,L]:
 *   den_c = eps[2p]*c_out + x;  den_u = eps[2p+1]*c_out + x;  den = den_u + guidance*(den_c - den_u);
 *   x = x + ((x - den) / sigma) * d_sigma + noise*sigma_up      (the branch `#define ORIGINAL_SAMPLER_ALGORITHMS 1`, samplers.h:66, selects, :1431-1449;
 *                                                                 sigma = sigma_i, d_sigma = sigma_down - sigma_i; noise may be NULL: term skipped)
 * every product and sum rounded separately to fp32, i.e. bit-identical to the host loop.  clip > 0 additionally clamps the new x to
 * [-clip, clip] (not in the reference; used with random-weight synthetic UNets, which do not denoise, to keep the trajectory finite). */
int osg_sampler_cfg_euler_a(osg_ctx* ctx, float* x, const float* eps, const float* noise, int prompts, long L, float c_out,
                            float guidance, float sigma, float d_sigma, float sigma_up, float clip);

/* ---- data movement ------------------------------------------------------------------------------------------ */
/* N-d transpose (XnnPack::transpose, onnxstream.cpp:1748): out.shape[i] = shape[perm[i]]. elem_size in {1,2,4,8}. */
int osg_transpose(osg_ctx* ctx, int elem_size, const void* x, void* y, int rank, const long* shape, const int* perm);
/* 2-D strided block copy: for o<outer: dst[o*dst_pitch + dst_off .. +inner) = src[o*src_pitch + src_off .. +inner)
 * (element units).  Implements Concat (:4140), Split (:5999), Slice (:6499) along any axis. */
int osg_copy_2d(osg_ctx* ctx, int elem_size, const void* src, long src_pitch, long src_off, void* dst, long dst_pitch,
                long dst_off, long outer, long inner);
/* Two-input Concat along the innermost run in ONE launch: dst[o][0:inner_a) = a[o][:], dst[o][inner_a:inner_a+inner_b) = b[o][:] for
 * o < outer; a, b dense (element units).  The UNet's 12 skip-connection concatenations (onnxstream.cpp:4140). */
int osg_concat2(osg_ctx* ctx, int elem_size, const void* a, long inner_a, const void* b, long inner_b, void* dst, long outer);
/* Nearest/asymmetric/floor resize of [N,H,W,C] (nhwc=1) or [N,C,H,W] (nhwc=0) by integer-or-not scales (onnxstream.cpp:6120-6315). */
int osg_resize_nearest(osg_ctx* ctx, int elem_size, const void* x, void* y, int N, int C, int H, int W, int Ho, int Wo, int nhwc);
/* Gather rows along axis 0: y[i,:] = x[idx[i],:] (onnxstream.cpp:6316). idx is a DEVICE int64 array. */
int osg_gather_rows(osg_ctx* ctx, int elem_size, const void* x, const int64_t* idx, void* y, long n_idx, long row_elems, long n_rows);
/* MaxPool NHWC (XnnPack::maxpool_nhwc, onnxstream.cpp:1536). */
int osg_maxpool_nhwc(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, int N, int H, int W, int C, int KH, int KW, int sh,
                     int sw, int pt, int pl, int pb, int pr);

/* ---- conversion / quantisation (XnnPack::convert :757, convert_qu8 :802, Model::dequantize :3353, quantize :3247) -- */
/* f16<->f32 ; u8->f32/f16: (float)((int)q - zp) * scale ; f32/f16->u8: clamp(rne(x * (1.0f/scale)) + zp, 0, 255). */
int osg_convert(osg_ctx* ctx, osg_dtype src_dtype, osg_dtype dst_dtype, const void* x, void* y, long n, float scale, int zero_point);

/* ---- uint8 arithmetic: the reference's m_use_uint8_arithmetic path (W8A8; the VAE decoder of `sd --rpi-lowmem`, src/sd.cpp:1212-1222) ------
 * A uint8 tensor is (codes, scale, zero_point); every entry point reproduces the reference's codes BIT FOR BIT (specification:
 * oracle/np_qu8.py, pinned against the reference's own intermediates).  Output (scale, zero_point) come from the caller: the host Model
 * derives them from range_data.txt with Model::range_to_scale (onnxstream.cpp:3234), exactly as the reference does per op (:4664-4687). */
/* XnnPack::convolution<uint8_t,int32_t> (onnxstream.cpp:1292, :1458-1491) + the bias rescaling of Model::run's Conv branch (:4639-4660):
 * acc = sum (x - x_zp)(w - w_zp) on v_mfma_i32_16x16x64_i8 (padding taps hold x_zp, i.e. contribute 0) + (int32)(bias / (x_scale * w_scale)),
 * then XNNPACK's fp32 requantisation  clamp(rne(float(acc) * (x_scale * w_scale / out_scale)))) + out_zp.  bias_f32: fp32 [Cout] or NULL. */
int osg_qu8_conv2d_nhwc(osg_ctx* ctx, const void* x, float x_scale, int x_zp, const void* w_ohwi, float w_scale, int w_zp, const float* bias_f32,
                        float out_scale, int out_zp, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w,
                        int pad_top, int pad_left, int pad_bottom, int pad_right);
/* The same with a caller-owned table.  The pipelined kernel streams both code matrices straight into LDS, where the descriptor's bounds check fills
 * the convolution halo with code 0 instead of x_zp; what every out-of-image tap then owes, x_zp * sum_c (w[n,tap,c] - w_zp), is settled in the
 * epilogue of the border rows from  tap_sums[Cout][KH*KW] = sum_c w[n,tap,c]  (int32).  The table depends on the weight alone:
 * osg_qu8_conv_tap_sums fills it (async, compute stream), a caller whose weight stays where it is fills it once.  tap_sums NULL (and
 * osg_qu8_conv2d_nhwc): the library rebuilds it in its workspace on every call. */
int osg_qu8_conv_tap_sums(osg_ctx* ctx, const void* w_ohwi, int Cout, int KH, int KW, int Cin, int* tap_sums);
int osg_qu8_conv2d_nhwc_t(osg_ctx* ctx, const void* x, float x_scale, int x_zp, const void* w_ohwi, float w_scale, int w_zp, const float* bias_f32,
                          float out_scale, int out_zp, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w,
                          int pad_top, int pad_left, int pad_bottom, int pad_right, const int* tap_sums);
/* XnnPack::matrix_multiply<uint8_t> (onnxstream.cpp:1035, Model::run MatMul uint8 branch :5779-5837): C[b] = requant(sum_k (A - a_zp)(B - b_zp)
 * (+ bias)), A:[M,K] rows lda apart, B given K-contiguous as [N,K]; stride_* = element strides between batch items (0 = shared). */
int osg_qu8_gemm(osg_ctx* ctx, const void* A, long lda, float a_scale, int a_zp, const void* B_nk, float b_scale, int b_zp, const float* bias_f32,
                 float out_scale, int out_zp, void* C, int M, int N, int K, int batch, long stride_a, long stride_b, long stride_c);
/* y[i] = lut[x[i]] with a 256-entry DEVICE table: Model::run's Sigmoid uint8 branch (onnxstream.cpp:4412-4481) is a function of the input code
 * alone -- the host builds the table with its own expf (the function the reference calls) and the lookup is exact by construction. */
int osg_qu8_lut(osg_ctx* ctx, const void* x, void* y, long n, const void* lut256);
/* XnnPack::add / multiply with quint8 parameters (onnxstream.cpp:1666 / :846; Model::run :5105-5124 / :3977-3996), NumPy broadcasting like osg_binary:
 * Mul = (a - a_zp)(b - b_zp) * (a_scale * b_scale / out_scale) in fp32, clamp, rne, + out_zp;  Add = XNNPACK's fixed-point add (20-bit multipliers). */
int osg_qu8_binary(osg_ctx* ctx, osg_binary_kind kind, const void* a, const long* a_shape, float a_scale, int a_zp, const void* b, const long* b_shape,
                   float b_scale, int b_zp, void* y, float out_scale, int out_zp, int rank);
/* Model::run's InstanceNormalization uint8 branch (onnxstream.cpp:4987-5043) on [rows, L]: dequantise, mean / variance / affine in double as the
 * reference does, requantise -- evaluated per distinct input code from the row's code histogram.  scale/bias: fp32 [n_scale] device vectors. */
int osg_qu8_instance_norm(osg_ctx* ctx, const void* x, void* y, int rows, long L, int n_scale, const float* scale, const float* bias, float eps,
                          float in_scale, int in_zp, float out_scale, int out_zp);
/* Mul(x, g[C]) -> Add(., b[C]) [-> Sigmoid -> Mul(., sigmoid)] of the uint8 graphs (GroupNorm affine + SiLU; reference multiply / add with quint8
 * parameters :3977-3996, :5105-5124, Sigmoid :4412-4481) as one pass: (scale, zero point) of x, g, the Mul's output (m_*), b, the Add's output (a_*),
 * the Sigmoid's output (s_*; sig_lut = its 256-entry table, NULL = no activation) and the final Mul's output (o_*).  Channel of element i:
 * (i / inner) % C.  The codes are those of the separate launches. */
int osg_qu8_affine_act(osg_ctx* ctx, const void* x, float x_scale, int x_zp, const void* g, float g_scale, int g_zp, float m_scale, int m_zp, const void* b,
                       float b_scale, int b_zp, float a_scale, int a_zp, const void* sig_lut, float s_scale, int s_zp, float o_scale, int o_zp, void* y, long n, int C,
                       long inner);
/* osg_qu8_instance_norm_nhwc followed by osg_qu8_affine_act with the normalisation's table lookup done inside the affine pass (n_out_* = the
 * normalisation's output parameters): histogram, tables, one pass.  Same codes as the separate ops. */
int osg_qu8_norm_affine_act_nhwc(osg_ctx* ctx, const void* x, long HW, int C, int G, int n_scale, const float* scale, const float* bias, float eps, float x_scale,
                                 int x_zp, float n_out_scale, int n_out_zp, const void* g, float g_scale, int g_zp, float m_scale, int m_zp, const void* b, float b_scale,
                                 int b_zp, float a_scale, int a_zp, const void* sig_lut, float s_scale, int s_zp, float o_scale, int o_zp, void* y);
/* ... the same op when the [1,G,L] view is a Reshape of an NHWC [HW][C] tensor (row g = channels [g*C/G, (g+1)*C/G) of every pixel): identical codes,
 * no layout copy around it. */
int osg_qu8_instance_norm_nhwc(osg_ctx* ctx, const void* x, void* y, long HW, int C, int G, int n_scale, const float* scale, const float* bias, float eps,
                               float in_scale, int in_zp, float out_scale, int out_zp);
/* XnnPack::softmax<uint8_t> (onnxstream.cpp:1958-2060 -> XNNPACK qu8 softmax): lut_u32_256 = DEVICE table t[i] = lrint(min(UINT32_MAX / C, 2^23 - 1) *
 * exp((i - 255) * in_scale)) built by the host; y = min(255, ((t[x + 255 - rowmax] << 8) + (sum >> 1)) / sum); output scale 1/256, zero point 0. */
int osg_qu8_softmax_last(osg_ctx* ctx, const void* x, void* y, long rows, long C, const void* lut_u32_256);

/* ---- developer probe (not part of the operator surface) ----------------------------------------------------------------------------------------
 * With OSG_KDBG=1 in the environment every contraction launch outside a graph capture records, per workgroup, 8 x int64 phase timestamps (100 MHz
 * s_memrealtime: 0 entry, 1 prologue loads issued, 2 first tile resident, 3 k loop done, 4 epilogue start, 5 epilogue stores issued, 6 stores
 * retired) into one device buffer that the next launch overwrites; this copies its first `bytes` to the host (tools/kernel_phase_probe.py). */
int osg_kdbg_read(osg_ctx* ctx, void* host_dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* OSGPU_H */
