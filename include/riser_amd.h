/*
 * riser_amd.h -- C ABI of the MI355X (gfx950) squiggle-classification hot path.
 *
 * This is the drop-in boundary for the path
 *     SignalProcessor.mad_normalise -> Model.classify -> ConvNet.forward -> softmax
 * of comprna/riser.  The reference has no native code and therefore no FFI of its own;
 * each entry point below names the reference Python interface it replaces
 * (file:line under /root/reference).  INTEGRATION.md shows the ctypes stub a RISER
 * maintainer would add to riser/model.py and riser/preprocess.py.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns RS_OK (0) or a
 *     negative rs_status and records a message readable with rs_last_error()
 *     (thread-local).
 *   - all `d_` pointers are DEVICE pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()); the library owns only the packed weight copies made by
 *     rs_model_create and frees them in rs_model_destroy.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all launches
 *     are asynchronous on it, nothing here synchronises the device.
 *   - raw signal samples are int16 ADC counts, as delivered by
 *     riser/client.py:47 (np.frombuffer(read.raw_data, signal_dtype)).
 *   - a batch is B reads; read b is the `d_len[b]` samples starting at element
 *     `d_off[b]` of `d_sig` (so a trim is just an offset: riser/preprocess.py:100,105).
 *   - `h_len` (rs_forward / rs_classify / rs_classify_ensemble / rs_autotune) is the HOST's copy of d_len, or NULL.
 *     With it the conv stack lays the batch out in packed blocks (each read occupies len / U + 1 blocks of
 *     U = rs_block_samples() samples, work proportional to every read's own length) - the host needs the block count to
 *     size the launches; without it every read takes the blocks of an Lmax-sample read.  Results are bit-identical
 *     either way.  It must mirror d_len: a device length that disagrees can only cost that read and the reads BEHIND it in
 *     the batch their results (a read whose blocks no longer fit the table the host sized is dropped: NaN probabilities),
 *     never a read in front of it, and never an access outside the workspace or the length array.
 *   - reads shorter than 2^n_layers samples (4096 for the shipped 12-layer net,
 *     riser/preprocess.py:8) cannot be classified: RS_ERR_LENGTH, matching the
 *     RuntimeError torch raises in max_pool1d for the reference.
 */
#ifndef RISER_AMD_H
#define RISER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#define RS_API __attribute__((visibility("default")))

typedef enum rs_status {
    RS_OK = 0,
    RS_ERR_ARG = -1,        /* null pointer, bad size, unsupported configuration */
    RS_ERR_HIP = -2,        /* a HIP runtime call failed; message has hipGetErrorString */
    RS_ERR_LENGTH = -3,     /* a read is empty / shorter than the network minimum / too long */
    RS_ERR_OOM = -4,        /* device or host allocation failed */
    RS_ERR_WORKSPACE = -5   /* caller's workspace is smaller than rs_workspace_bytes() */
} rs_status;

/* arithmetic type of the conv stack (accumulation is always fp32) */
typedef enum rs_dtype {
    RS_F32 = 0,             /* f32-input MFMA (v_mfma_f32_16x16x4_f32): exact fmaf chains */
    RS_BF16 = 1,            /* bf16 activations/weights, v_mfma_f32_16x16x32_bf16, fp32 accumulate */
    RS_F16 = 2,             /* f16 activations/weights, v_mfma_f32_16x16x32_f16, fp32 accumulate */
    RS_F32W = 3,            /* fp32 throughout, conv lowered to Winograd F(2,3) / F(4,3) on the f32-input MFMA:
                               2/3 resp. 1/2 of the multiplications of RS_F32, probabilities within ~1e-5 of it */
    RS_BF16X3 = 4,          /* split precision on the bf16 MFMA: every activation and weight is a pair hi + lo of bf16
                               values (~16 significand bits), a product is hi*hi + lo*hi + hi*lo on three
                               v_mfma_f32_16x16x32_bf16 with fp32 accumulate.  The 16-bit mode that meets the 1e-3
                               probability tolerance of BASELINE configs 3 / 5; RS_BF16 / RS_F16 are the fast,
                               approximate variants */
    RS_F16X3 = 5,           /* the same with f16 pairs (~22 significand bits above 2^-14, less below) */
    RS_F16XF8 = 6           /* RS_F16X3 whose wide layers keep hi*hi on v_mfma_f32_16x16x32_f16 and evaluate the two cross
                               terms hi*lo + lo*hi as ONE K-concatenated product of OCP e4m3 values on the block-scaled
                               v_mfma_scale_f32_16x16x128_f8f6f4 (an E8M0 scale per activation row and 32 channels): two
                               instruction times per 64 K where split precision spends three; probabilities within ~3e-4
                               of fp32 (inside the 1e-3 tolerance), same activation range as RS_F16X3 */
} rs_dtype;

/* decisions of riser/control.py:75-82, as written into rs_decide's output */
typedef enum rs_decision {
    RS_TRY_AGAIN = 0,
    RS_ACCEPT = 1,
    RS_REJECT = 2,
    RS_NO_DECISION = 3
} rs_decision;

typedef enum rs_mode { RS_ENRICH = 0, RS_DEPLETE = 1 } rs_mode;

typedef struct rs_model rs_model;

/* Message of the last failing call on this thread ("" if none). */
RS_API const char* rs_last_error(void);

/* ABI version of this header: (major << 16) | minor. */
RS_API int rs_version(void);

/* Number of HIP devices visible (0 if none / no driver).  Does not select a device. */
RS_API int rs_device_count(void);

/*
 * Build a model on `device` from a reference-format ConvNet state dict.
 * Replaces Model.__init__ (riser/model.py:7-20) + ConvNet.__init__ (riser/nets/cnn.py:8-41)
 * for the shipped configuration: depth 1, kernel 3, classifier `gap_fc`.
 *   n_layers            number of conv blocks (config.cnn.n_layers, 12 in every shipped yaml)
 *   channels[n_layers]  config.cnn.channels
 *   conv_w[i]           HOST fp32 [channels[i], channels[i-1] (1 for i==0), 3]  = layers.{i}.0.weight
 *   conv_b[i]           HOST fp32 [channels[i]]                                   = layers.{i}.0.bias
 *   fc_w                HOST fp32 [n_classes, channels[n_layers-1]]               = classifier.2.weight
 *   fc_b                HOST fp32 [n_classes]                                     = classifier.2.bias
 *   n_classes           must be 2 (riser/control.py:69 unpacks exactly two probabilities)
 * Host pointers need only live for the duration of the call.
 */
RS_API int rs_model_create(int n_layers, const int32_t* channels, int n_classes,
                    const float* const* conv_w, const float* const* conv_b,
                    const float* fc_w, const float* fc_b,
                    int dtype /* rs_dtype */, int device, rs_model** out);

RS_API int rs_model_destroy(rs_model* m);

/*
 * ABI 2.2.  Replace the model's gap_fc head by the reference's `fc` classifier (riser/nets/cnn.py:22-27):
 *     Flatten(1) -> Linear(C * positions, hidden) -> ReLU -> Linear(hidden, 2),
 * C = the last conv layer's channels.  w1 [hidden][C * positions] (row-major, features channel-major as Flatten orders
 * them: f = c * positions + p), b1 [hidden], w2 [2][hidden], b2 [2]: host pointers, copied.  The reference hard-codes
 * C * positions = 67 * 753 and hidden = 4096 (one input length, 12048 .. 12063 samples, of one 4-layer net); here any
 * positions >= 1 and any hidden that is a multiple of 64.  fp32 models only (RS_F32 / RS_F32W).  After this call a read
 * whose len >> n_layers differs from `positions` cannot be classified - the reference's matmul raises for it
 * (riser/nets/cnn.py:47) - and gets NaN probabilities (decision RS_NO_DECISION); rs_workspace_bytes() grows by the
 * partial sums of the first Linear.  The fc model runs in rs_forward / rs_classify / rs_classify_ensemble like any other
 * (ensemble members must agree on `hidden`).  Call once, before the first forward.
 */
RS_API int rs_model_set_fc_classifier(rs_model* m, int positions, int hidden, const float* w1, const float* b1,
                               const float* w2, const float* b2);

/* Bytes of device workspace rs_forward / rs_classify need for a batch of B reads of at
 * most Lmax samples (0 on bad arguments): the block table, the normalised signals and two activation buffers. */
RS_API size_t rs_workspace_bytes(const rs_model* m, int B, int Lmax);

/* Largest B one call accepts for reads of up to Lmax samples: every activation buffer is addressed through a 2 GiB
 * buffer-resource window (32-bit offsets, hardware bounds checking).  Reads are independent: split bigger batches. */
RS_API int rs_max_batch(const rs_model* m, int Lmax);

/* Block size U of the packed activation layout in samples: 2^max(n_layers, 12) (4096 for the shipped net; doubled when
 * RS_WINO4 puts the last layer on the F(4,3) lowering).  Read b of a batch occupies len_b / U + 1 blocks.  Since ABI 2.1 the
 * layout has TWO LEVELS: this is the block of the last three conv layers and of the head; the layers before them (and the
 * normalised signal) run on finer blocks - rs_layer_info.block_samples, 1024 samples for the shipped net - so that a read
 * costs its own length more closely (a live 8615-sample read: 9 x 1024 instead of 3 x 4096 samples of rows); the library
 * re-packs the (small) buffer in between.  In conv layer `layer`'s output buffer read b starts at row
 * (Uf * sum_{i<b}(len_i / Uf + 1)) >> (layer + 1), Uf = that layer's block_samples.  Results do not depend on the layout. */
RS_API int rs_block_samples(const rs_model* m);

/*
 * MAD normalisation + outlier smoothing of B reads.
 * Replaces SignalProcessor.mad_normalise (riser/preprocess.py:108-147), bit-exact in
 * float64: exact integer median / MAD, y = (x - med) / (1.4826 * mad), then the
 * sequential in-place smoothing of |y| > 3.5 (order-dependent recurrence, un-clipped
 * ends).  mad == 0 gives zeros (the reference returns an int64 zero array).
 *   d_out32  fp32 [B, ld32] or NULL: (float)y, i.e. the cast of riser/model.py:25;
 *            elements [len, pad_to) of each row are zero-filled (pad_to <= ld32)
 *   d_out64  fp64 [B, ld64] or NULL: y exactly as the reference returns it
 *   d_stats  fp64 [B, 2] or NULL: (median, mad)
 * Lmax is the host-known maximum of d_len (sizes the LDS staging buffer); every read
 * must have 1 <= len <= Lmax <= 65536.
 */
RS_API int rs_normalise(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B, int Lmax,
                 float* d_out32, int64_t ld32, int32_t pad_to,
                 double* d_out64, int64_t ld64, double* d_stats, void* stream);

/*
 * The same for FLOATING-POINT signals (elem_bytes = 2: float16 (since ABI 2.2), 4: float32, 8: float64), the other input
 * SignalProcessor.mad_normalise accepts (riser/preprocess.py:108-115; the retrain path normalises pA-scaled float
 * signals, riser/retrain/preprocess.py:79).  numpy keeps the input's precision end to end (float32 in -> float32
 * arithmetic -> float32 out, the Python constant 1.4826 adopting the array's dtype under NEP 50), and so does this
 * entry: d_out has the element type of d_sig, rows of `ld` elements, bit-identical to the reference's result (float16:
 * every operation as numpy evaluates it - in float32, rounded to half - and the median's mean of the two middle values
 * with its float32 accumulator).
 * Every read must have 1 <= len; read b is d_sig[d_off[b] .. d_off[b] + d_len[b]) in elements.  d_stats: fp64 [B, 2]
 * = (median, mad) or NULL.  Off the live path: an exact radix select per read, not tuned.
 */
RS_API int rs_normalise_float(const void* d_sig, int elem_bytes, const int64_t* d_off, const int32_t* d_len, int B,
                       void* d_out, int64_t ld, double* d_stats, void* stream);

/*
 * Forward pass + softmax on already normalised fp32 signals.
 * Replaces Model.classify (riser/model.py:22-28) -> ConvNet.forward
 * (riser/nets/cnn.py:43-65) for a batch: B independent reads of individual length.
 *   d_x       fp32 [B, ldx]; row b holds d_len[b] samples (what follows them is not read)
 *   h_len     host copy of d_len or NULL (see the conventions above)
 *   d_probs   fp32 [B, 2] = (p_off_target, p_on_target), the order of riser/control.py:69
 *   d_logits  fp32 [B, 2] or NULL
 *   Lmin      host-known lower bound of d_len (0 if unknown; ignored when h_len is given).  Only a speed hint:
 *             tiles that fall entirely into the padding behind a read are skipped; when Lmin says no such tile
 *             can exist the per-tile test is not even compiled into the walk.
 */
RS_API int rs_forward(rs_model* m, const float* d_x, int64_t ldx, const int32_t* d_len, const int32_t* h_len, int B, int Lmin,
               int Lmax, void* d_ws, size_t ws_bytes, float* d_probs, float* d_logits, void* stream);

/* Samples a read of Lmax samples occupies in the packed layout: (Lmax / U + 1) * U, U = rs_block_samples(). */
RS_API int rs_padded_length(const rs_model* m, int Lmax);

/*
 * Fused path: raw int16 reads -> normalise -> forward -> probabilities.
 * Equivalent to the pair of calls at riser/control.py:63 and :69 for every read of the
 * batch; the normalised signals live in the workspace.
 */
RS_API int rs_classify(rs_model* m, const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, const int32_t* h_len,
                int B, int Lmin, int Lmax, void* d_ws, size_t ws_bytes,
                float* d_probs, float* d_logits, void* stream);

/*
 * Optional tile-shape tuning for one batch geometry (not part of the reference; like a BLAS "find" mode).
 * Runs rs_classify on the given batch and, for every conv layer that runs a tiled kernel, times each feasible
 * entry of that kernel's tile-shape table in place (HIP events on `stream`, which is synchronised repeatedly);
 * a shape more than 3 % faster than the launch planner's choice is remembered for launches with the same number of
 * rows (blocks x block length) of that layer.  Results are bit-identical whatever shape runs (the accumulation order
 * over channels does not depend on the tile shape; 16-bit: on the panel width it does, within fp32 round-off).
 * d_probs receives the batch's probabilities as rs_classify would produce them; *n_changed (optional) the number
 * of layers whose choice changed.  Costs a few hundred launches: call once per deployment batch size, not per batch.
 */
RS_API int rs_autotune(rs_model* m, const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, const int32_t* h_len, int B,
                int Lmin, int Lmax, void* d_ws, size_t ws_bytes, float* d_probs, int32_t* n_changed, void* stream);

/*
 * Ensemble form of rs_classify: the model loop of riser/control.py:68-71 for a whole batch.
 * The reads are normalised ONCE (riser/control.py:63), every model of `models` (same architecture,
 * e.g. the mRNA / mtRNA / globin stand-ins of one kit) runs its forward pass on the same normalised
 * signals, and - if d_decision is not NULL - the decision of riser/control.py:75-82 is taken on the
 * device (see rs_decide for `max_len`, `threshold`, `mode`).
 *   d_probs     fp32 [n_models, B, 2]
 *   d_decision  uint8 [B] or NULL
 * The workspace is shared by the models (block table and normalised signals once).  With
 * rs_ensemble_workspace_bytes(models, n_models, B, Lmax) bytes every model has its own pair of activation buffers and the
 * forwards run CONCURRENTLY (model 0 on `stream`, the others on library-owned side streams forked behind the normalise
 * launch and joined in front of the decision; everything the call enqueues is ordered on `stream` as before) whenever the
 * batch under-fills the chip (fewer than ~1800 blocks of 4096 samples: a live ReadUntil batch; larger batches fill it by
 * themselves and run back to back).  With less,
 * but at least the MAXIMUM of rs_workspace_bytes(models[k], B, Lmax) over the models, they run back to back on `stream`;
 * below that the call is refused with RS_ERR_WORKSPACE.  The probabilities are the same bits either way.
 */
RS_API size_t rs_ensemble_workspace_bytes(rs_model* const* models, int n_models, int B, int Lmax);
RS_API int rs_classify_ensemble(rs_model* const* models, int n_models, const int16_t* d_sig, const int64_t* d_off,
                         const int32_t* d_len, const int32_t* h_len, int B, int Lmin, int Lmax, void* d_ws,
                         size_t ws_bytes, float* d_probs, uint8_t* d_decision, int max_len, float threshold, int mode,
                         void* stream);

/*
 * Ensemble decision of riser/control.py:75-82 for B reads and n_models models.
 *   d_probs   fp32 [n_models, B, 2]
 *   max_len   SignalProcessor.get_max_length() (riser/preprocess.py:36-40)
 *   d_out     uint8 [B], rs_decision values
 * Comparisons are strict `>` in fp32, as torch does for `tensor > python_float`.
 */
RS_API int rs_decide(const float* d_probs, int n_models, int B, const int32_t* d_len, int max_len,
              float threshold, int mode /* rs_mode */, uint8_t* d_out, void* stream);

/*
 * poly(A) end detector on raw reads.
 * Replaces SignalProcessor.get_polyA_end (riser/preprocess.py:42-79): 500-sample
 * windows, start when the window mean rose > 20 % over the previous 1000 samples with
 * window MAD <= 20, end at the first later window with MAD > 20.
 *   d_end     int32 [B]: the end index (a multiple of 500) or -1 where the reference
 *             returns None
 * Reads may have any length (the whole signal is scanned, as the reference does; the scan stops at the first end
 * found, which nothing later in the signal can change).
 */
RS_API int rs_polya_end(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B,
                 int32_t* d_end, void* stream);

/*
 * ABI 2.2.  The same detector, RESUMABLE: for a caller that sees a read again, longer, with an unchanged prefix (a read that
 * stays in its pore between two ReadUntil batches; riser/control.py:53 re-scans it from its first sample each time).
 * The detector's state after W whole 500-sample windows without an end - W, the `start` index, the sums of the last two
 * windows - depends on the read's first 500 W samples only:
 *   d_state_in   int32 [B][4] (windows done, start, sum of window W - 2, sum of window W - 1) as a previous call returned
 *                it for the SAME read, all zeros for a read seen for the first time, or NULL (= all zeros for every read);
 *                a state whose window count exceeds the read's is ignored (the read is scanned whole)
 *   d_state_out  int32 [B][4]: the state after this scan; for a read whose end is found, its input state unchanged
 *                (resumed from it the scan finds the same end).  Must not be d_state_in.
 * d_end as rs_polya_end: the results are identical to a scan from the first sample.  The caller answers for the prefix being
 * unchanged (riser_amd's control loop: the signal store's verified delta path).
 */
RS_API int rs_polya_end_resume(const int16_t* d_sig, const int64_t* d_off, const int32_t* d_len, int B,
                        const int32_t* d_state_in, int32_t* d_end, int32_t* d_state_out, void* stream);

/*
 * Scatter n segments of raw samples between device buffers: segment k = d_src[d_src_off[k] .. + d_len[k]) is copied to
 * d_dst[d_dst_off[k] ..) (offsets in samples).  The batched control loop keeps the raw signal of every read in flight
 * resident on the device: an AccumulatingCache client (riser/client.py:29-31) re-sends a read's whole signal with every
 * batch, and only the samples that are new since the last batch are uploaded (compacted, one transfer) and scattered
 * behind the ones already there.
 */
RS_API int rs_copy_segments(const int16_t* d_src, int16_t* d_dst, const int64_t* d_src_off, const int64_t* d_dst_off,
                     const int32_t* d_len, int n, void* stream);

/* Introspection used by bench.py for roofline accounting: per conv layer i (0-based),
 * fills the padded GEMM shape the kernels execute.  Returns RS_ERR_ARG if out of range. */
typedef struct rs_layer_info {
    int32_t c_in, c_out;        /* logical channels */
    int32_t cp_in, cp_out;      /* padded row widths of the activation buffers */
    int32_t k_pad;              /* padded reduction length (3 * padded input channels) */
    int32_t n_pad;              /* padded output channels the MFMA tiles cover */
    int32_t bm, bn, kc;         /* workgroup tile (rows x couts) and channel chunk of the last launch (0 if never run) */
    int32_t gemm_row_div;       /* conv rows per GEMM row: 1 direct lowering, 2 Winograd F(2,3), 4 Winograd F(4,3) */
    int32_t block_samples;      /* ABI 2.1: block size (samples) of the packed layout this layer runs on - its input rows and
                                   its output rows; the output of the last fine layer is then re-packed to coarse blocks */
    int32_t rows_format;        /* ABI 2.5: layout of the layer's OUTPUT rows: 0 one value per channel (fp32 / 16-bit), 1 split
                                   precision (panels [hi x 32 | lo x 32]), 2 F8 rows (RS_F16XF8: per 64 channels hi16 x 64, then
                                   [hi8 | lo8 | hi8 | lo8] x 32 e4m3 bytes; E8M0 scale plane behind the rows) */
} rs_layer_info;
RS_API int rs_model_layer_info(const rs_model* m, int layer, rs_layer_info* out);

/*
 * Sequential conv programs: the secondary ResNet architecture (riser/nets/resnet.py:7-131; not
 * loadable by the reference's Model, no shipped config or weights).  The host folds eval-mode
 * BatchNorm into each conv and hands over an execution list; activations live in `n_buffers`
 * numbered buffers (0 = the normalised input [B, L], the rest carved from the workspace).
 * One uniform length per batch.  Convs run on the f32-input MFMA (generic over k / stride / pad, not tuned per shape);
 * also used for ConvNet configurations outside the shipped class (depth > 1, kernels other than 3).
 */
typedef struct rs_seq_op {
    int32_t kind;           /* 0 = conv1d (+bias, +residual, +relu), 1 = MaxPool1d(2, 2, padding `pad` = 0 or 1) */
    int32_t src, dst, add;  /* buffer ids; add = -1 for no residual input */
    int32_t c_in, c_out, k, stride, pad, relu;
    const float* w;         /* HOST fp32 [c_out, c_in, k], BN already folded in (conv only) */
    const float* b;         /* HOST fp32 [c_out] */
} rs_seq_op;
typedef struct rs_seqnet rs_seqnet;
RS_API int rs_seqnet_create(const rs_seq_op* ops, int n_ops, int n_buffers, const float* fc_w /* [2, c_last] */,
                     const float* fc_b, int c_last, int device, rs_seqnet** out);
RS_API int rs_seqnet_destroy(rs_seqnet* m);
RS_API size_t rs_seqnet_workspace_bytes(const rs_seqnet* m, int B, int L);
RS_API int rs_seqnet_forward(rs_seqnet* m, const float* d_x /* fp32 [B, L] */, int B, int L, void* d_ws, size_t ws_bytes,
                      float* d_probs, float* d_logits, void* stream);
/*
 * ABI 2.3: arithmetic of a program's stem and residual basic blocks (riser/nets/resnet.py:50-57, shortcuts :21-24,45-47; the
 * bottleneck blocks :60-70 too when RS_SEQ_BNECK_X3 was set at rs_seqnet_create - measured slower than their fp32 form, so off):
 * where a ResNet's time is.  RS_F32 (default: the f32-input MFMA) or RS_BF16X3: split precision
 * on the bf16 MFMA as RS_BF16X3 of rs_model_create computes it (hi + lo pairs, three v_mfma_f32_16x16x32_bf16 per product, fp32
 * accumulate; activations stay fp32 between launches) - BASELINE.json's "1D-ResNet forward pass ... MFMA bf16" inside the 1e-3
 * tolerance.  The head and unfused ops keep fp32.  RS_ERR_ARG for any other dtype, or for a program without a fused residual block.
 */
RS_API int rs_seqnet_set_mode(rs_seqnet* m, int dtype /* rs_dtype */);
/*
 * ABI 2.4: RAGGED batches - the reads of a ReadUntil batch have their own lengths (riser/control.py:36-60: anything from the
 * minimum to the kit's maximum), and rs_seqnet_forward takes one length per call.  Here read b is d_x[b * ld .. b * ld + d_len[b]):
 * the rows of every read after every op are computed on the device from d_len, every kernel masks by them, the buffers keep the
 * row pitches of an ld-sample read (rs_seqnet_workspace_bytes(m, B, ld)).  A read's probabilities are those of a uniform call on
 * it alone, bit for bit.  Only for programs whose ops all run inside fused launches (stem + residual blocks: what
 * riser/nets/resnet.py builds): rs_seqnet_ragged_ok(m) == 1; otherwise RS_ERR_ARG (group the reads by length instead).
 */
/* Preconditions of rs_seqnet_forward_ragged: d_len[b] in [0, ld] (a larger value is read as ld); a read shorter than the network
 * needs (no output row left behind some op) gets NaN probabilities - a defined result, no out-of-range access; a batch whose
 * buffers outgrow the kernels' 2 GiB windows returns RS_ERR_ARG ("split the batch") before anything wrong is written:
 * rs_seqnet_max_batch(m, ld) is the largest B that cannot. */
RS_API int rs_seqnet_max_batch(const rs_seqnet* m, int L);
RS_API int rs_seqnet_ragged_ok(const rs_seqnet* m);
RS_API int rs_seqnet_forward_ragged(rs_seqnet* m, const float* d_x /* fp32 [B, ld] */, const int32_t* d_len, int B, int ld, void* d_ws,
                             size_t ws_bytes, float* d_probs, float* d_logits, void* stream);

/* Half precision has a range: RS_F16 / RS_F16X3 / RS_F16XF8 store activations as IEEE half, and a value beyond 65504 leaves the
 * conversion as +inf - the forward pass goes on, the probabilities of that read are wrong, and the reference's fp32 path
 * (riser/model.py:22-28) has no such failure.  Every kernel epilogue of those modes checks its conversions and raises a sticky flag
 * on the model; the launch entry points still return RS_OK.  rs_model_saturated waits for `stream` and returns 1 if any call
 * since the last reset overflowed, 0 if none did (always 0 for the fp32 and bf16 modes), a negative rs_status on error; `reset`
 * != 0 clears the flag behind the read.  (ABI 2.5) */
RS_API int rs_model_saturated(rs_model* m, int reset, void* stream);

/*
 * Test hook: every following forward pass of `m` also copies the output buffer of conv layer `layer`
 * (1 <= layer < n_layers; position-major [NB * (U >> (layer + 1)), cp_out] in the packed block layout - see
 * rs_block_samples - fp32 or 16-bit) into d_dst
 * (at most `bytes`).  d_dst = NULL switches it off.  Used by the layer-wise parity tests.
 */
RS_API int rs_debug_capture_layer(rs_model* m, int layer, void* d_dst, size_t bytes);

/*
 * Stage timing with HIP events on the launch stream (used by bench.py's roofline leg).
 * While enabled, rs_forward / rs_classify record one event before their first launch and
 * one after every kernel launch, on the caller's stream; no synchronisation is added.
 * rs_profile_read synchronises on the last recorded event and ADDS the elapsed
 * milliseconds per stage into stage_ms[0 .. n_layers + 1]:
 *   stage 0 = normalise, stage 1 = layer-0 conv, stage 1 + i = conv layer i (i >= 1),
 *   stage n_layers + 1 = GAP/FC/softmax head;
 * *calls receives the number of profiled forward calls.  Recorded events are consumed.
 * on = 1: one event per launch (an event costs ~4.5 us of stream time: 14 per call).  on = 2: coarse - events only
 * at the call's start, after normalise + layer 0 (their time lands in stage 1), after the LAST conv layer (the
 * whole conv stack, layers 1 .. n-1, lands in stage n_layers) and after the head: what bench.py's timed region uses.
 */
RS_API int rs_profile_enable(rs_model* m, int on);
RS_API int rs_profile_read(rs_model* m, float* stage_ms, int32_t* calls);

#ifdef __cplusplus
}
#endif
#endif /* RISER_AMD_H */
