/*
 * t2mi.h -- C-ABI of libt2mi.so: the MI355X (gfx950) Tacotron2 mel-generation hot path.
 *
 * The reference (nii-yamagishilab/tacotron2) has no FFI layer of its own: its "operator API" for
 * this path is the Python class protocol of tacotron/tacotron_v2.py, tacotron/rnn_wrappers.py and
 * tacotron/helpers.py, executed by TensorFlow 1.x.  Each entry point below replaces the TensorFlow
 * execution of the reference symbols it cites (paths relative to the reference root).  The Python
 * package `tacotron2_amd` binds these with ctypes and mirrors the reference classes on top.
 *
 * Conventions
 *  - every function returns T2_OK (0) or a negative T2_E* code and never throws across the ABI;
 *    t2_last_error() returns a thread-local message for the last failure on this thread.
 *  - "dev" pointers are HIP device pointers on the ctx's device (e.g. torch.Tensor.data_ptr());
 *    "host" pointers are ordinary host memory.  The caller owns every I/O buffer; the library owns
 *    weights and workspaces inside its handles.
 *  - all floating point is IEEE fp32; ids / lengths are int64 (util/tfrecord.py:108-109).
 *  - a handle is not thread-safe; use one ctx per GPU per thread.  All work is enqueued on the
 *    ctx's stream.  Host synchronisation: t2_decode_free() / t2_decode_free_finish() (to return
 *    steps_out); and every call that runs a persistent kernel (below) waits for it once, to read
 *    its error word: t2_encode*, t2_decode_teacher, t2_train_* at the paper-size dimension
 *    class.  Everything else (state set-up, t2_postnet, t2_decode_step, the launch-per-phase
 *    schedules of other sizes) only enqueues.
 *  - Persistent kernels.  At the paper-size dimension class the decoder loop, the encoder BiLSTM
 *    and the training recurrences each run as ONE launch whose workgroups hand data to each other
 *    (csrc/persist.h, persist_bwd.h).  That needs all of them resident at once, i.e. the whole
 *    GPU.  (a) Within a process the library serialises such launches per GPU on the device
 *    (stream-wait on the previous one's completion event), whatever streams / threads / engines
 *    issue them.  (b) Every wait inside them is bounded (250 ms); if another PROCESS holds part
 *    of the GPU and a wait runs out, the kernel raises an error word, and the call that reads it
 *    re-runs the same work on the launch-per-phase schedule before it returns: results are
 *    always valid, T2_OK is never returned over invalid buffers.  The event is counted
 *    (t2_model_get_stat "persist_timeouts") and the model's next 16 decodes skip the persistent
 *    kernels.  For full speed give the process the GPU to itself.
 *  - batch size B <= T2_MAX_BATCH per call (one reference batch: the stop rule of
 *    helpers.py:90-94 couples all rows of a batch, so a batch is never split internally).
 */
#ifndef T2MI_H
#define T2MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2_ABI_VERSION 1
#define T2_MAX_BATCH 64
#define T2_MAX_PRENET_LAYERS 4

#define T2_OK 0
#define T2_EINVAL (-1)  /* bad argument / unsupported size  -> Python ValueError / AssertionError   */
#define T2_ENOMEM (-2)  /* hipMalloc failed                                                          */
#define T2_EHIP (-3)    /* a HIP runtime call failed (message in t2_last_error)                      */
#define T2_ESTATE (-4)  /* call order violated (model not finalized, weight missing, ...)            */
#define T2_ERANGE (-5)  /* embedding id outside [0, num_symbols): modules.py:65-66 assert ops        */

typedef struct t2_ctx t2_ctx;
typedef struct t2_model t2_model;
typedef struct t2_state t2_state;

/* Every dimension of the path is a runtime parameter.  Field meaning:
 * hparams.py:14,23,27,28,50-52 and the constructor arguments of tacotron_v2.py:35-36,83-91,149-151,183. */
typedef struct t2_dims {
    int32_t num_symbols;          /* Embedding rows, hparams.py:23                                  */
    int32_t embedding_dim;        /* Embedding width                                                */
    int32_t enc_conv_layers;      /* EncoderV2 num_conv_layers                                      */
    int32_t enc_kernel;           /* EncoderV2 kernel_size                                          */
    int32_t enc_units;            /* EncoderV2 out_units (even; BiLSTM has enc_units/2 per dir)     */
    int32_t num_prenet_layers;    /* len(prenets) of AttentionRNN, <= T2_MAX_PRENET_LAYERS          */
    int32_t prenet_units[T2_MAX_PRENET_LAYERS];
    int32_t attention_lstm_units; /* A: cell of AttentionRNN (ZoneoutLSTMCell)                      */
    int32_t decoder_lstm_units;   /* D: DecoderRNNV2 out_units                                      */
    int32_t attention_dim;        /* U: LocationSensitiveAttention num_units                        */
    int32_t attention_filters;    /* F: hparams.py:51                                               */
    int32_t attention_kernel;     /* K: hparams.py:50                                               */
    int32_t cumulative_weights;   /* hparams.py:52                                                  */
    int32_t num_mels;             /* M: hparams.py:14                                               */
    int32_t outputs_per_step;     /* r: hparams.py:27                                               */
    int32_t n_feed_frame;         /* nf <= r: hparams.py:28, helpers.py:55                          */
    int32_t postnet_layers;       /* PostNetV2 num_postnet_layers                                   */
    int32_t postnet_kernel;       /* PostNetV2 kernel_size                                          */
    int32_t postnet_channels;     /* PostNetV2 out_channels                                         */
    float zoneout_cell;           /* ZoneoutLSTMCell zoneout_factor_cell, modules.py:237            */
    float zoneout_output;         /* ZoneoutLSTMCell zoneout_factor_output                          */
    float bn_eps;                 /* tf.layers.batch_normalization epsilon (1e-3)                   */
    float forget_bias;            /* tf.nn.rnn_cell.LSTMCell forget_bias (1.0)                      */
} t2_dims;

int t2_version(void);
const char* t2_last_error(void);

/* One context per GPU per thread.  hip_stream may be NULL (the library creates its own stream). */
int t2_ctx_create(int device, void* hip_stream, t2_ctx** out);
int t2_ctx_destroy(t2_ctx* ctx);
int t2_ctx_synchronize(t2_ctx* ctx);

/* Model = dims + named fp32 weights in TensorFlow layouts (Dense kernel [in,out]; LSTM kernel
 * [in+h,4h], rows [inputs;h_prev], column blocks i|j|f|o; conv kernel [k,in,out]; BN vectors [C]).
 * Variable names: tacotron2_amd/weights.py:weight_shapes.  Validation mirrors the reference's
 * constructor checks: enc_units even (tacotron_v2.py:40), nf <= r (helpers.py:55), zoneout factors
 * in [0,1] (modules.py:241-245) -> T2_EINVAL. */
int t2_model_create(t2_ctx* ctx, const t2_dims* dims, t2_model** out);
int t2_model_set_weight(t2_model* m, const char* name, const float* host_ptr, const int64_t* shape, int rank);
int t2_model_finalize(t2_model* m); /* repack into kernel-native layouts and upload */
int t2_model_destroy(t2_model* m);

/* Embedding.call (modules.py:64-67) + EncoderV2.call (tacotron_v2.py:53-63), inference arithmetic.
 * ids host int64 [B,T_in]; lens host int64 [B] or NULL (no sequence_length); memory_out dev fp32
 * [B,T_in,enc_units].  T2_ERANGE if an id is outside [0,num_symbols). */
int t2_encode(t2_model* m, const int64_t* ids, const int64_t* lens, int B, int T_in, float* memory_out);

/* EncoderV2.call alone on a dev fp32 [B,T_in,embedding_dim] input (tacotron_v2.py:53-63). */
int t2_encode_embedded(t2_model* m, const float* emb, const int64_t* lens, int B, int T_in, float* memory_out);

/* PostNetV2.call (tacotron_v2.py:200-204): mel dev [B,T_out,M] -> mel_post dev [B,T_out,M]. */
int t2_postnet(t2_model* m, const float* mel, int B, int T_out, float* mel_post);

/* Stand-alone layers / cells of the reference's protocol (API parity: the decode loops never go through these).
 * t2_dense: tf.layers.Dense as PreNet.call uses it without dropout (modules.py:88-91) -- y[rows][N] = act(x[rows][K] . kernel[K][N]
 * + bias[N]); activation 0 none, 1 relu, 2 tanh; bias may be NULL.  All pointers dev.
 * t2_zoneout_lstm_cell: ZoneoutLSTMCell.__call__ with is_training = False (modules.py:261-283 over tf.nn.rnn_cell.LSTMCell,
 * rnn_impl.py:25-33): xh dev [B][K] = concat(inputs, h_prev), kernel dev [K][4H] (column blocks i | j | f | o), bias dev [4H];
 * z_ws dev [B][4H] workspace; outputs h_raw (the cell's output), c_new / h_new (the zoned state), dev [B][H]. */
int t2_dense(t2_ctx* ctx, const float* x, const float* kernel, const float* bias, int rows, int K, int N, int activation, float* y);
int t2_zoneout_lstm_cell(t2_ctx* ctx, const float* xh, const float* c_prev, const float* h_prev, const float* kernel, const float* bias,
                         int B, int K, int H, float zoneout_cell, float zoneout_output, float forget_bias, float* z_ws, float* h_raw,
                         float* c_new, float* h_new);
/* The sampled (is_training = True) forms of the same two layers, masks from the library's counter-based generator
 * u = u01[seed, stream, t, row, column] (csrc/kernels.h; TensorFlow's stateful random stream cannot be reproduced, the oracle shares it):
 * t2_dropout: tf.layers.dropout(training=True) as PreNet.call applies it (modules.py:90,96-98) -- y = x / (1 - rate) where u >= rate, else 0;
 * t2_zoneout_lstm_cell_train: ZoneoutLSTMCell.__call__ with is_training = True (modules.py:274-276) -- c' = (1 - zc) dropout(c_new - c) + c,
 * h' likewise with stream + 1; h_raw is the cell output. */
int t2_dropout(t2_ctx* ctx, const float* x, long long rows, int cols, float rate, uint32_t seed, int stream, int t, float* y);
int t2_zoneout_lstm_cell_train(t2_ctx* ctx, const float* xh, const float* c_prev, const float* h_prev, const float* kernel, const float* bias,
                               int B, int K, int H, float zoneout_cell, float zoneout_output, float forget_bias, uint32_t seed, int stream, int t,
                               float* z_ws, float* h_raw, float* c_new, float* h_new);

/* Decoder state = OutputAndStopTokenWrapper.zero_state (rnn_wrappers.py:99-100) of the cell stack
 * OutputAndStopTokenWrapper(DecoderRNNV2(AttentionRNN(ZoneoutLSTMCell, prenets,
 * LocationSensitiveAttention))) plus the attention mechanism's constructor side effects
 * (tacotron_v2.py:95-101 -> BahdanauAttention: values = memory masked by lens, keys = values .
 * memory_layer).  memory dev [B,T_in,enc_units]; lens host int64 [B] or NULL. */
int t2_state_create(t2_model* m, const float* memory, const int64_t* lens, int B, int T_in, t2_state** out);
int t2_state_reset(t2_state* s);    /* back to zero_state, time = 0 */
int t2_state_destroy(t2_state* s);
/* Copy one state tensor to a dev buffer (row-major fp32).  names: "attention_c", "attention_h",
 * "lstm1_c", "lstm1_h", "lstm2_c", "lstm2_h" ([B,units]), "attention" ([B,enc_units]),
 * "alignments", "attention_state" ([B,T_in]), "keys" ([B,T_in,U]). */
int t2_state_get(t2_state* s, const char* name, float* out);
/* further names: "attention_output" ([B,A]), "lstm1_output", "decoder_output" ([B,D]): the raw cell outputs of the LAST step, i.e.
 * what DecoderPreNetWrapper.call (rnn_wrappers.py:51-53), the first ZoneoutLSTM of DecoderRNNV2 and DecoderRNNV2.call
 * (tacotron_v2.py:177-178) return. */

/* RNNCell.call parity: one OutputAndStopTokenWrapper.call (rnn_wrappers.py:105-109) =
 * prenets (rnn_wrappers.py:51-53) -> AttentionWrapper step with LocationSensitiveAttention.__call__
 * (tacotron_v2.py:118-141) -> concat (rnn_wrappers.py:76-78) -> 2 ZoneoutLSTM (tacotron_v2.py:155-161)
 * -> mel / stop projections.  prev_frames dev [B, M*nf]; mel_r dev [B, M*r]; stop dev [B] (logit).
 * The state is advanced in place. */
int t2_decode_step(t2_model* m, t2_state* s, const float* prev_frames, float* mel_r, float* stop);

/* The same cell call with the reference's training-mode sampling: DecoderPreNetWrapper.call (rnn_wrappers.py:51-53) with the
 * PreNets' dropout active (modules.py:90,96-98: is_training or apply_dropout_on_inference; rate prenet_drop_rate, 0 = off)
 * and, when sample_zoneout != 0, ZoneoutLSTMCell's sampled branch (modules.py:274-276) in the three cells; masks come from
 * the library's counter-based generator at (seed, stream, time = the state's step count, row, unit), i.e. the masks
 * t2_train_decoder_forward draws at that step with the same seed (TensorFlow's random stream is not reproducible). */
int t2_decode_step_train(t2_model* m, t2_state* s, const float* prev_frames, float prenet_drop_rate, int sample_zoneout,
                         uint32_t seed, float* mel_r, float* stop);

/* dynamic_decode(BasicDecoder(cell, TrainingHelper | ValidationHelper, zero_state)) as wired by
 * DecoderV1.call (tacotron_v1.py:134-152) for the V2 cell.  targets dev [B,T_out,M] (T_out % r == 0),
 * S = T_out / r steps for every row (helpers.py:154-156, :105,:133).
 * teacher_forcing != 0: next input = targets[:, t, -M*nf:] (helpers.py:182 / :134);
 * teacher_forcing == 0: next input = own output[:, -M*nf:] (ValidationHelper, helpers.py:135-136).
 * Outputs (dev): mel [B,T_out,M]; stop [B,S] (logits); align [S,B,T_in] (alignment_history.stack()). */
int t2_decode_teacher(t2_model* m, const float* memory, const int64_t* lens, const float* targets,
                      int B, int T_in, int T_out, int teacher_forcing,
                      float* mel, float* stop, float* align);

/* dynamic_decode(BasicDecoder(cell, StopTokenBasedInferenceHelper, zero_state), maximum_iterations)
 * (tacotron_v1.py:143-150; helpers.py:83-94): free-running decode, finished when every row's stop
 * logit is > 0 at a step with time > min_iters (batch-coupled), or at max_iters; the output of the
 * finishing step is kept.  Output buffers (dev) are sized for max_iters steps:
 * mel [B, max_iters*r, M]; stop [B, max_iters]; align [max_iters, B, T_in]; only the first
 * *steps_out steps are written.  steps_out: host int. */
int t2_decode_free(t2_model* m, const float* memory, const int64_t* lens, int B, int T_in,
                   int max_iters, int min_iters, float* mel, float* stop, float* align, int* steps_out);

/* final_state of dynamic_decode as DecoderV1.call returns it (tacotron_v1.py:149-153; models.py:53 reads
 * final_state[0].alignment_history): one tensor of the cell stack's state after the LAST decode loop of this model
 * (t2_decode_teacher / t2_decode_free).  Names and shapes as t2_state_get. */
int t2_decode_final_state(t2_model* m, const char* name, float* out);

/* Same decode, split for callers that time or overlap it: _launch enqueues everything on the
 * ctx stream without synchronising, _finish waits and returns the step count. */
int t2_decode_free_launch(t2_model* m, const float* memory, const int64_t* lens, int B, int T_in,
                          int max_iters, int min_iters, float* mel, float* stop, float* align);
int t2_decode_free_finish(t2_model* m, int* steps_out);

/* Tuning / measurement hooks (no reference counterpart).  Options: "graph" (1: decode loops replay
 * hipGraph chunks, 0: eager launches), "chunk" (decoder steps per graph, 1..64), "splitk" (1: split-K LSTM step for batches > 16), "hoist" (1, default: the recurrent halves of the LSTM gate GEMMs ride as extra workgroups in the attention / projection launches; 0: whole-K LSTM launches), "att3" (1, default: batches of <= 16 rows put all three riders into the attention launch), "lstm_only" (1: decode steps launch only the
 * LSTM kernels -- timing runs, outputs meaningless), "time_lstm" (1:
 * record HIP events around every decoder LSTM launch in eager mode; resets the counters).
 * Stats: "lstm_launches", "lstm_ms" (sum of event-timed launch durations, synchronises). */
int t2_model_set_option(t2_model* m, const char* name, int value);
int t2_model_get_stat(t2_model* m, const char* name, double* out);

/* ---- training-step math (BASELINE.json configs[3]) ---------------------------
 * All buffers are dev fp32.  out3 = {sum of weighted losses, number of non-zero broadcast weights,
 * sum / count (0 when count == 0)}: tf.losses.compute_weighted_loss with its default
 * SUM_BY_NONZERO_WEIGHTS reduction.  The first two let a data-parallel caller all-reduce numerator and
 * count over ranks and divide afterwards, which reproduces the single-device value of the whole batch. */

/* SingleSpeakerTacotronV1Model.spec_loss (tacotron/models.py:136-144): |y_hat - y| [B,T,M] weighted by
 * mask [B,T] broadcast over M (priority-frequency blend unused: defaults None / 0). */
int t2_spec_loss(t2_ctx* ctx, const float* y_hat, const float* y, const float* mask, int B, int T, int M, float* out3);

/* Allocation-free variant of t2_spec_loss.  It has no counterpart in the reference; it exists for callers that cannot
 * allocate (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): it calls no hipMalloc /
 * hipFree or allocating routine.  Same arithmetic and results as t2_spec_loss; the per-block partial sums live in a 16 KiB
 * device array of the library (one copy per GPU, held until the process ends; its address is looked up in t2_ctx_create).
 * Limit: B*T*M <= 2^28 elements, else T2_EINVAL.  Calls that overlap on one GPU (two streams / two contexts) race on that
 * array: serialise them. */
int t2_spec_loss_noalloc(t2_ctx* ctx, const float* y_hat, const float* y, const float* mask, int B, int T, int M, float* out3);

/* binary_loss (tacotron/models.py:146-148): tf.losses.sigmoid_cross_entropy(done, logits, weights = mask),
 * logits / done / mask [B,S]. */
int t2_binary_loss(t2_ctx* ctx, const float* logits, const float* done, const float* mask, int B, int S, float* out3);

/* The optimizer block of model_fn (tacotron/models.py:67-76) over ONE flat parameter / gradient buffer of
 * n_params floats (16-byte aligned): tf.clip_by_global_norm(grads, clip_norm) followed by
 * tf.train.AdamOptimizer(lr, beta1, beta2, epsilon).apply_gradients -- TF arithmetic: epsilon outside the
 * square root, bias correction folded into lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t).
 * grad_scale multiplies the gradients first (1 / world_size after a sum all-reduce); clip_norm <= 0
 * disables clipping.  The library owns m, v and the step count t. */
typedef struct t2_optim t2_optim;
int t2_optim_create(t2_ctx* ctx, int64_t n_params, float beta1, float beta2, float epsilon, t2_optim** out);
int t2_optim_destroy(t2_optim* o);
int t2_optim_global_norm(t2_optim* o, const float* grads, float grad_scale, float* norm_out /* dev [1] */);
int t2_optim_step(t2_optim* o, float* params, const float* grads, float learning_rate, float clip_norm, float grad_scale);
/* checkpoint / resume of the slots (NULL pointers are skipped) */
int t2_optim_get_state(t2_optim* o, float* m_out, float* v_out, int64_t* step_out);
int t2_optim_set_state(t2_optim* o, const float* m_in, const float* v_in, int64_t step);

/* ---- training step of the V2 path (BASELINE.json configs[3]) ---------------------------------------------------------
 * A trainer owns the tape of one (B, T_in, S = T_out / r) teacher-forced training forward and the workspaces of the
 * backward pass.  Sampled paths (dropout, zoneout) draw from the counter-based generator of csrc/kernels.h (function t2_u01) keyed by
 * `seed`: TensorFlow's stateful random streams cannot be reproduced, the arithmetic around the masks is TF's. */
typedef struct t2_trainer t2_trainer;
int t2_trainer_create(t2_model* m, int B, int T_in, int S, t2_trainer** out);
int t2_trainer_destroy(t2_trainer* tr);
/* DecoderV1.call(is_training=True) with the V2 cells (tacotron_v1.py:134-150): TrainingHelper (helpers.py:141-184), PreNet
 * dropout with prenet_drop_rate (modules.py:90,96-98), ZoneoutLSTMCell training branch (modules.py:274-276).  Buffers and
 * outputs as t2_decode_teacher; targets dev [B, S*r, M]. */
int t2_train_decoder_forward(t2_trainer* tr, const float* memory, const int64_t* lens, const float* targets,
                             float prenet_drop_rate, uint32_t seed, float* mel, float* stop, float* align);

/* Embedding.call + EncoderV2.call with is_training = True (tacotron_v2.py:53-63; Conv1d.call modules.py:149-157): every conv
 * layer normalises with the batch statistics of all B*T_in positions (padding included), then relu, then dropout(drop_rate);
 * the BiLSTM runs ZoneoutLSTMCell's training branch with sequence lengths.  ids host int64 [B,T_in]; memory_out dev
 * [B,T_in,enc_units]; bn_stats_out dev (may be NULL): per conv layer [batch mean | biased batch variance], 2*enc_units floats
 * each -- the moving-statistics update of the step (models.py:75) is moving = 0.99*moving + 0.01*batch on the caller's side. */
int t2_train_encoder_forward(t2_trainer* tr, const int64_t* ids, const int64_t* lens, float drop_rate, uint32_t seed,
                             float* memory_out, float* bn_stats_out);
/* EncoderV2.call(is_training = True) alone on a dev fp32 [B,T_in,embedding_dim] input (the Layer the reference defines,
 * tacotron_v2.py:53-63); same arithmetic and outputs as t2_train_encoder_forward after its embedding lookup.  No backward through
 * the embedding is defined for this entry point. */
int t2_train_encoder_forward_embedded(t2_trainer* tr, const float* emb, const int64_t* lens, float drop_rate, uint32_t seed,
                                      float* memory_out, float* bn_stats_out);

/* PostNetV2.call with is_training=True (tacotron/tacotron_v2.py:181-204 over Conv1d.call, tacotron/modules.py:149-157):
 * batch-statistics batch normalisation over all B*T_out positions, tanh (none on the last convolution), dropout from the
 * library's counter-based generator, then Dense(out_units) + residual.  Forward only (the reference's loss, models.py:58-61,
 * does not contain the PostNet).  bn_stats_out: optional dev [postnet_layers][2][postnet_channels] (mean | biased variance). */
int t2_train_postnet_forward(t2_model* m, const float* mel, int B, int T_out, float drop_rate, uint32_t seed, float* mel_post,
                             float* bn_stats_out);
/* Backward of t2_train_encoder_forward: dmemory dev [B,T_in,enc_units] (from t2_train_decoder_backward) -> gradients of
 * `embedding`, encoder/conv1d_i/{kernel, batch_normalization/gamma, beta} (through the batch statistics) and the BiLSTM
 * kernels / biases into flat_grads at the bound offsets. */
int t2_train_encoder_backward(t2_trainer* tr, const float* dmemory, float* flat_grads);
/* After the optimizer step: rebuild every device copy of the weights (packed operands, folded batch norms, the transposed
 * operands of the backward pass) from the flat parameter buffer (dev; layout = the bound offsets, TF layouts).  The inference
 * entry points of the same model see the new weights too. */
int t2_trainer_load_params(t2_trainer* tr, const float* flat_params);
/* The UPDATE_OPS of the step (tacotron/models.py:75): moving = momentum*moving + (1-momentum)*batch for the encoder's batch
 * norms (variance with Bessel's correction), in place in the flat parameter buffer; statistics of the last encoder forward. */
int t2_trainer_update_moving_stats(t2_trainer* tr, float* flat_params, float momentum);
/* Data-parallel training (no reference counterpart: the reference is single-device).  counts dev [2] = number of
 * non-zero loss weights of the WHOLE batch over all ranks for {spec_loss, binary_loss} (the caller all-reduces them and keeps the
 * buffer alive); the next backward passes then differentiate sum_rank / count_total, so that the plain SUM of the ranks' gradients
 * is the gradient of tf.losses' SUM_BY_NONZERO_WEIGHTS value of the whole batch (models.py:136-148).  NULL: this rank's own counts. */
int t2_trainer_set_loss_counts(t2_trainer* tr, const float* counts);
/* Where the gradient of variable `name` (TF layout) goes in the caller's flat gradient buffer (floats, multiple of 4). */
int t2_trainer_bind_gradient(t2_trainer* tr, const char* name, int64_t offset);
/* Backward of t2_train_decoder_forward through the losses of model_fn (tacotron/models.py:58-61: spec_loss(mel) +
 * binary_loss(stop), T1 / T2) -- BPTT over the tape.  done, bin_mask dev [B,S]; spec_mask dev [B,S*r].  Writes the gradient of
 * every decoder/ and attention/ variable into flat_grads (dev) at the bound offsets, d loss / d memory into dmemory
 * (dev [B,T_in,enc_units]) and {mel_loss, done_loss, loss} into loss_out (dev [3]). */
int t2_train_decoder_backward(t2_trainer* tr, const float* done, const float* spec_mask, const float* bin_mask,
                              float* flat_grads, float* dmemory, float* loss_out);

/* ---- Tacotron v1 path (BASELINE.json configs[4]) ---------------------------------
 * Field meaning: reference hparams.py:14-15,23,27-28,31-47,54-60.  Inference arithmetic (dropout off, batch
 * norm with moving statistics).  Variable names: tacotron2_amd/weights_v1.py:weight_shapes_v1 (TF GRUCell
 * layouts: gates/kernel [in+h, 2h] = r | u, candidate/kernel [in+h, h]). */
typedef struct t2v1_dims {
    int32_t num_symbols, embedding_dim;
    int32_t enc_prenet_units[2];       /* EncoderV1 prenet_out_units (tacotron_v1.py:25); [0] == embedding_dim      */
    int32_t cbhg_out_units, conv_channels, max_filter_width, proj1_channels, proj2_channels, num_highway;
    int32_t dec_prenet_units[2];       /* DecoderV1 prenet_out_units (tacotron_v1.py:96)                             */
    int32_t attention_out_units;       /* GRU units = BahdanauAttention num_units (tacotron_v1.py:58-59), <= 256     */
    int32_t decoder_out_units;         /* OutputProjectionWrapper size = residual GRU units (tacotron_v1.py:70-74)    */
    int32_t num_mels, outputs_per_step, n_feed_frame;
    int32_t post_cbhg_out_units, post_conv_channels, post_max_filter_width, post_proj1_channels, post_proj2_channels,
            post_num_highway, num_freq;
    float bn_eps;
} t2v1_dims;
typedef struct t2v1_model t2v1_model;

/* Validation mirrors the reference's checks: nf <= r (helpers.py:55), even CBHG out_units (modules.py:169),
 * prenet_out_units[0] == embedding width (tacotron_v1.py:41-44), output_size >= 1 (rnn_wrappers.py:155-156). */
int t2v1_model_create(t2_ctx* ctx, const t2v1_dims* dims, t2v1_model** out);
int t2v1_model_set_weight(t2v1_model* m, const char* name, const float* host_ptr, const int64_t* shape, int rank);
int t2v1_model_finalize(t2v1_model* m);
int t2v1_model_destroy(t2v1_model* m);
int t2v1_model_set_option(t2v1_model* m, const char* name, int value);     /* "graph", "chunk", "gruseq", "tfbatch"; trainers: "skinny",
                                                                               "train_persist" (1: each decoder-trainer recurrence is one
                                                                               persistent launch), "inject_timeout" (tests) */
/* counters: "train_persist_launches", "persist_timeouts" (as t2_model_get_stat; no counterpart in the reference) */
int t2v1_model_get_stat(t2v1_model* m, const char* name, double* out);

/* Embedding.call + EncoderV1.call (tacotron_v1.py:46-49): PreNets -> CBHG.  lens NULL = no sequence_length, which
 * is how the V1 model_fn calls it (models.py:47).  memory_out dev [B,T_in,cbhg_out_units]. */
int t2v1_encode(t2v1_model* m, const int64_t* ids, const int64_t* lens, int B, int T_in, float* memory_out);
int t2v1_encode_embedded(t2v1_model* m, const float* emb, const int64_t* lens, int B, int T_in, float* memory_out);

/* CBHG.call (modules.py:205-229) alone: which = 0 the encoder's CBHG (x dev [B,T,enc_prenet_units[1]]),
 * 1 the PostNet's (x dev [B,T,num_mels]); out dev [B,T,out_units]. */
int t2v1_cbhg(t2v1_model* m, int which, const float* x, const int64_t* lens, int B, int T, float* out);

/* PostNet.call (tacotron_v1.py:180-183): mel dev [B,T_out,num_mels] -> linear spectrogram dev [B,T_out,num_freq]. */
int t2v1_postnet(t2v1_model* m, const float* mel, int B, int T_out, float* spec_out);

/* DecoderV1.call (tacotron_v1.py:118-153): OutputAndStopTokenWrapper(DecoderRNNV1(OutputProjectionWrapper(
 * AttentionRNN(GRUCell, PreNets, BahdanauAttention)), 2 x ResidualWrapper(GRUCell))) under dynamic_decode with
 * TrainingHelper / ValidationHelper (_teacher) or StopTokenBasedInferenceHelper (_free).  Arguments and outputs as
 * t2_decode_teacher / t2_decode_free. */
int t2v1_decode_teacher(t2v1_model* m, const float* memory, const int64_t* lens, const float* targets, int B, int T_in,
                        int T_out, int teacher_forcing, float* mel, float* stop, float* align);
int t2v1_decode_free(t2v1_model* m, const float* memory, const int64_t* lens, int B, int T_in, int max_iters, int min_iters,
                     float* mel, float* stop, float* align, int* steps_out);

/* ---- training step of the Tacotron-1 PostNet model (BASELINE.json configs[4] "train_postnet") -------
 * TacotronV1PostNetModel.model_fn(mode=TRAIN) (tacotron/models.py:190-233): PostNet(is_training=True) (tacotron_v1.py:156-186)
 * = CBHG with batch-statistics batch norm (modules.py:163-232, Conv1d.call :149-157) -> Dense(num_freq); the masked L1 loss
 * (models.py:208-211, :265-274); the backward pass of all of it.  The optimizer block is t2_optim_* above; train_postnet.py:30-72
 * is the caller.  The trainer reads every v1/postnet/ variable from ONE flat fp32 device buffer (TF layouts, where
 * t2v1_postnet_trainer_bind says) and writes gradients to a buffer of the same layout, so an optimizer step on that buffer needs
 * no refresh.  Built for post_net_cbhg_out_units == 256 (the published hparams; EINVAL otherwise).  One trainer per (B, T). */
typedef struct t2v1_ptrainer t2v1_ptrainer;
int t2v1_postnet_trainer_create(t2v1_model* m, int B, int T, t2v1_ptrainer** out);
int t2v1_postnet_trainer_destroy(t2v1_ptrainer* tr);
/* variable `name` starts `offset` floats into the flat parameter / gradient buffers (a multiple of 4) */
int t2v1_postnet_trainer_bind(t2v1_ptrainer* tr, const char* name, int64_t offset);
/* PostNet(is_training=True)(mel) (tacotron_v1.py:180-183; the CBHG gets no lengths): mel dev [B,T,num_mels], flat_params dev,
 * spec_out dev [B,T,num_freq].  Leaves the tape of the backward pass and the batch statistics in the trainer. */
int t2v1_postnet_train_forward(t2v1_ptrainer* tr, const float* flat_params, const float* mel, float* spec_out);
/* CBHG.call(is_training=True) (tacotron/modules.py:205-229) as a block: the CBHG of the trainer (a PostNet trainer: v1/postnet/cbhg,
 * an encoder trainer: v1/encoder/cbhg) with batch-statistics batch norm.  x dev [B][T][in_units], lens host [B] or NULL,
 * out dev [B][T][out_units].  No backward pass follows this entry. */
int t2v1_cbhg_train_forward(t2v1_ptrainer* tr, const float* flat_params, const float* x, const int64_t* lens, float* out);
/* Gradient of spec_loss(last forward, spec_target dev [B,T,num_freq], mask dev [B,T]) into flat_grads (dev; every trainable
 * v1/postnet/ slice is written, nothing else); loss_out dev [3] = {sum, non-zero weight count, loss}.  glob_count: dev [1], the
 * count of a larger data-parallel batch, or NULL (this batch's own).  Consumes the tape: ESTATE without a forward before it. */
int t2v1_postnet_train_backward(t2v1_ptrainer* tr, const float* spec_target, const float* mask, const float* glob_count,
                                float* flat_grads, float* loss_out);
/* tf.layers.batch_normalization's UPDATE_OPS (models.py:229-232) of the ten batch norms, in place in flat_params, from the batch
 * statistics of the last forward (Bessel-corrected variance). */
int t2v1_postnet_trainer_update_moving_stats(t2v1_ptrainer* tr, float* flat_params, float momentum);
/* Rebuild the model's inference copies of the PostNet (t2v1_postnet; model_fn EVAL / PREDICT, models.py:236-259) from the flat
 * buffer, on the device. */
int t2v1_postnet_trainer_export(t2v1_ptrainer* tr, const float* flat_params);

/* EncoderV1(is_training=True) with the embedding in front (tacotron_v1.py:18-49, modules.py:80-91, models.py:40-47): the encoder
 * half of SingleSpeakerTacotronV1Model's training step (the decoder half: t2v1_train_decoder_* below).
 * Same trainer type, flat-buffer convention, bind / update_moving_stats / export / destroy calls as the PostNet trainer above
 * (variables: embedding, v1/encoder/...).  ids host int64 [B,T_in] (T2_ERANGE outside the table); lens host int64 [B] or NULL
 * (models.py:46-47 passes none); PreNet dropout masks from the counter-based generator (seed); memory_out dev
 * [B,T_in,cbhg_out_units].  _backward: dmemory dev = d loss / d memory -> gradients of every encoder variable and the table. */
int t2v1_encoder_trainer_create(t2v1_model* m, int B, int T_in, t2v1_ptrainer** out);
int t2v1_encoder_train_forward(t2v1_ptrainer* tr, const float* flat_params, const int64_t* ids, const int64_t* lens, float drop_rate,
                               uint32_t seed, float* memory_out);
int t2v1_encoder_train_forward_embedded(t2v1_ptrainer* tr, const float* flat_params, const float* emb, const int64_t* lens,
                                        float drop_rate, uint32_t seed, float* memory_out);   /* EncoderV1.call on an embedded input */
/* demb_out: dev [B,T_in,embedding_dim] or NULL, the gradient of the embedded input; the table's gradient is written after a
 * forward from ids only */
int t2v1_encoder_train_backward(t2v1_ptrainer* tr, const float* dmemory, float* flat_grads, float* demb_out);

/* DecoderV1.call(is_training=True) (tacotron_v1.py:94-153: PreNets with dropout, AttentionRNN(GRUCell, BahdanauAttention),
 * OutputProjectionWrapper, two ResidualWrapper(GRUCell), mel / stop projections, under TrainingHelper helpers.py:141-184) with a
 * tape, and the backward pass of loss = spec_loss(mel) + binary_loss(stop) (models.py:58-61): with the encoder trainer above,
 * SingleSpeakerTacotronV1Model.model_fn(mode=TRAIN) (models.py:40-76).  One trainer per (B, T_in, S = T_out / r).  Flat-buffer
 * convention as above over the v1/decoder/ and v1/attention/ variables.  memory dev [B,T_in,cbhg_out_units]; lens host [B] or
 * NULL (memory_sequence_length); target dev [B,S*r,num_mels]; mel_out dev [B,S*r,num_mels], stop_out dev [B,S], align_out dev
 * [S,B,T_in].  _backward: done / bin_mask dev [B,S], spec_mask dev [B,S*r] -> gradients into flat_grads, d loss / d memory into
 * dmemory (dev [B,T_in,cbhg_out_units]), {mel_loss, done_loss, loss} into loss_out (dev [3]).  _set_loss_counts: dev [2], the
 * non-zero loss-weight counts of a larger data-parallel batch (NULL: this batch's own). */
typedef struct t2v1_dtrainer t2v1_dtrainer;
int t2v1_decoder_trainer_create(t2v1_model* m, int B, int T_in, int S, t2v1_dtrainer** out);
int t2v1_decoder_trainer_destroy(t2v1_dtrainer* tr);
int t2v1_decoder_trainer_bind(t2v1_dtrainer* tr, const char* name, int64_t offset);
int t2v1_decoder_trainer_set_loss_counts(t2v1_dtrainer* tr, const float* counts);
int t2v1_train_decoder_forward(t2v1_dtrainer* tr, const float* flat_params, const float* memory, const int64_t* lens, const float* target,
                               float drop_rate, uint32_t seed, float* mel_out, float* stop_out, float* align_out);
int t2v1_train_decoder_backward(t2v1_dtrainer* tr, const float* done, const float* spec_mask, const float* bin_mask, float* flat_grads,
                                float* dmemory, float* loss_out);

/* ---- spectrogram -> waveform ------------------------------------------------------------
 * Audio.inv_spectrogram_tf (util/audio.py:55-59): inv_amp_tf (:51-53: clip to [0,1], de-normalise with min_level_db,
 * + ref_level_db, dB -> amplitude), ** power, then _griffin_lim_tf (:66-82): y = inverse_stft(S); `iters` times
 * { est = stft(y); y = inverse_stft(S * est / max(1e-8, |est|)) } with tf.contrib.signal.stft / inverse_stft at
 * (frame_length = win_length, frame_step = hop_length, fft_length = 2 (num_freq - 1)), periodic Hann windows
 * (_stft_tf / _istft_tf / _stft_parameters :88-100).  spectrogram dev [B,T,num_freq] (normalised, as the V1 PostNet
 * emits it), wav dev [B, (T - 1) * hop_length + win_length].  The reference runs B = 1 (synthesize.py:39); rows are
 * independent.  n_fft must be a power of two <= 4096 (the reference: 2048). */
int t2_griffin_lim(t2_ctx* ctx, const float* spectrogram, int B, int T, int num_freq, int win_length, int hop_length,
                   float min_level_db, float ref_level_db, float power, int iters, float* wav);

#ifdef __cplusplus
}
#endif
#endif /* T2MI_H */
