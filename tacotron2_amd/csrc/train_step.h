// Training step of the V2 path (BASELINE.json configs[3]): the teacher-forced decoder forward in training-mode
// arithmetic (PreNet dropout, sampled zoneout) that records a tape, and the backward pass over it.
// Included by t2mi.hip.
#pragma once

struct t2_trainer {
    t2_model* m = nullptr;
    int B = 0, T = 0, S = 0;
    void* slab = nullptr;
    size_t slab_bytes = 0;
    TrainTape tape{};
    t2_state* state = nullptr;          // decoder state + staging owned by the trainer (never evicted by other decodes)
    bool have_forward = false;
};

extern "C" int t2_trainer_destroy(t2_trainer* tr) {
    if (!tr) return T2_OK;
    (void)hipSetDevice(tr->m->ctx->device);
    (void)hipStreamSynchronize(tr->m->ctx->stream);
    if (tr->slab) (void)hipFree(tr->slab);
    if (tr->state) state_free(tr->state);
    delete tr;
    return T2_OK;
}

extern "C" int t2_trainer_create(t2_model* m, int B, int T_in, int S, t2_trainer** out) {
    if (!m || !out) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    T2_TRY(check_batch(B, T_in));
    if (S < 1) return fail(T2_EINVAL, "S must be >= 1");
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    t2_trainer* tr = new t2_trainer();
    tr->m = m; tr->B = B; tr->T = T_in; tr->S = S;
    const int H[3] = {d.attention_lstm_units, d.decoder_lstm_units, d.decoder_lstm_units};
    size_t off = 0;
    auto carve = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4, 256); return o; };
    const size_t SB = (size_t)S * B;
    size_t o_xp[T2_MAX_PRENET_LAYERS], o_z[3], o_cp[3], o_cn[3], o_hr[3], o_hs[3];
    for (int i = 0; i < d.num_prenet_layers; ++i) o_xp[i] = carve(SB * d.prenet_units[i]);
    for (int l = 0; l < 3; ++l) {
        o_z[l] = carve(SB * 4 * H[l]); o_cp[l] = carve(SB * H[l]); o_cn[l] = carve(SB * H[l]);
        o_hr[l] = carve(SB * H[l]); o_hs[l] = carve(SB * H[l]);
    }
    const size_t o_q = carve(SB * d.attention_dim), o_sp = carve(SB * T_in), o_ctx = carve(SB * d.enc_units);
    tr->slab_bytes = off;
    hipError_t e = hipMalloc(&tr->slab, off);
    if (e != hipSuccess) { delete tr; return fail(T2_ENOMEM, "hipMalloc(trainer tape, %zu bytes) failed: %s", off, hipGetErrorString(e)); }
    char* p = (char*)tr->slab;
    for (int i = 0; i < d.num_prenet_layers; ++i) tr->tape.xp[i] = (float*)(p + o_xp[i]);
    for (int l = 0; l < 3; ++l) {
        tr->tape.z[l] = (float*)(p + o_z[l]); tr->tape.cprev[l] = (float*)(p + o_cp[l]); tr->tape.cnew[l] = (float*)(p + o_cn[l]);
        tr->tape.hraw[l] = (float*)(p + o_hr[l]); tr->tape.hstate[l] = (float*)(p + o_hs[l]);
    }
    tr->tape.q = (float*)(p + o_q); tr->tape.state_prev = (float*)(p + o_sp); tr->tape.ctx = (float*)(p + o_ctx);
    if (attn_lds_bytes(m->d, T_in) > 120 * 1024) { t2_trainer_destroy(tr); return fail(T2_EINVAL, "T_in %d too long for the attention kernel's LDS", T_in); }
    int rc = state_alloc(m, B, T_in, S, true, &tr->state);
    if (rc != T2_OK) { t2_trainer_destroy(tr); return rc; }
    *out = tr;
    return T2_OK;
}

// DecoderV1.call(is_training=True) with the V2 cells (tacotron_v1.py:134-150): TrainingHelper, S = T_out / r steps for every
// row, PreNet dropout with prenet_drop_rate (modules.py:90,96-98), ZoneoutLSTMCell training branch (modules.py:274-276).
// Outputs as t2_decode_teacher; the tape stays in the trainer for t2_train_decoder_backward.
extern "C" int t2_train_decoder_forward(t2_trainer* tr, const float* memory, const int64_t* lens, const float* targets,
                                        float prenet_drop_rate, uint32_t seed, float* mel, float* stop, float* align) {
    if (!tr || !memory || !targets || !mel || !stop || !align) return fail(T2_EINVAL, "null argument");
    if (!(prenet_drop_rate >= 0.f && prenet_drop_rate < 1.f)) return fail(T2_EINVAL, "drop rate must be in [0, 1)");
    t2_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = tr->B, T_in = tr->T, S = tr->S, M = d.num_mels, r = d.outputs_per_step;
    t2_state* s = tr->state;
    T2_TRY(state_init(s, memory, lens));
    T2_TRY(state_zero(s, S, 0, 0));
    HIP_TRY(hipMemcpyAsync(s->i_targets, targets, (size_t)B * S * r * M * 4, hipMemcpyDeviceToDevice, st));
    tr->tape.seed = seed; tr->tape.prenet_drop_rate = prenet_drop_rate;
    StepIO io = loop_io(s, 0);
    io.train = &tr->tape;
    io.slot_hint = 0;
    for (int t = 0; t < S; ++t) {                  // eager launches: seed and rates are per-call kernel arguments
        T2_TRY(launch_step(s, 0, io));
        hipLaunchKernelGGL(advance_ctl_kernel, dim3(1), dim3(64), 0, st, s->ctl, 1);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mel, s->o_mel, (size_t)B * S * r * M * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(stop, s->o_stop, (size_t)B * S * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(align, s->o_hist, (size_t)S * B * T_in * 4, hipMemcpyDeviceToDevice, st));
    tr->have_forward = true;
    return T2_OK;
}
