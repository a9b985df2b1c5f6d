/* sonicsim_hip.h -- C-ABI of libsonicsim_hip.so (MI355X / gfx950 moving-source audio renderer).
 *
 * The reference (JusperLee/SonicSim) has no FFI layer: its boundary for this path is the set of
 * module-level Python functions that SonicSim-SonicSet/SonicSet.py imports (SonicSet.py:16-21).
 * Every entry point below names the reference function (file:line, relative to the reference
 * repository root) whose arithmetic it replaces; the ctypes stubs a maintainer adds on the
 * reference side are shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / C++ types.
 *   - every function returns 0 on success or a negative SS_E* code; ss_last_error() returns a
 *     thread-local human-readable message.  No exception crosses the boundary.
 *   - flags bit 0 (SS_FLAG_DEVICE_PTR): data pointers are DEVICE pointers (zero copy; work is
 *     enqueued on `stream` and the call does not synchronise).  Otherwise they are HOST pointers:
 *     the library stages H2D/D2H itself and returns after the result is in host memory.
 *   - `stream` is a hipStream_t (NULL = the default stream).
 *   - the caller owns every buffer; inputs are never modified unless documented.
 *   - layouts are C-contiguous float32 unless stated: x[T], rirs[P][C][L], y[C][T].
 */
#ifndef SONICSIM_HIP_H
#define SONICSIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_VERSION 100 /* 0.1.0 */

#define SS_OK 0
#define SS_EINVAL (-1)   /* bad argument (shape, range, NULL) -> Python ValueError  */
#define SS_EHIP (-2)     /* HIP runtime error                  -> Python RuntimeError */
#define SS_ENOMEM (-3)   /* device/host allocation failed      -> Python MemoryError  */
#define SS_ENODEV (-4)   /* no usable GPU                      -> Python RuntimeError */

#define SS_FLAG_DEVICE_PTR 0x1u   /* data pointers are device pointers                       */
#define SS_FLAG_PATH_OS 0x10u     /* force the overlap-save (transform) engine               */
#define SS_FLAG_PATH_DIRECT 0x20u /* force the direct-form engine (exact fp32 fmaf chain)    */
#define SS_FLAG_GEOM_2048 0x40u   /* overlap-save engine: force B = 2048 / 256-thread workgroups (two parity passes) */
#define SS_FLAG_GEOM_4096 0x80u   /* overlap-save engine: force B = 4096 / 512-thread persistent workgroups          */
#define SS_FLAG_GEOM_13 0x200u    /* overlap-save engine: force the software-pipelined B = 4096 geometry (tvfir13.h; default for long filters) */
#define SS_FLAG_GEOM_ASM 0x400u   /* overlap-save engine: force the hand-scheduled gfx950 assembly kernel (segment / fixed schedules) */
#define SS_FLAG_LAYOUT_TC 0x100u  /* audio is [T][C] instead of [C][T] (loudness / mix calls) */
#define SS_FLAG_META_DEVICE 0x1000u /* ss_rir_bank_synth*_f32 with SS_FLAG_DEVICE_PTR: SsRirParams.delay / .dgain are DEVICE pointers (no staging copy) */
#define SS_FLAG_RESULT_DEVICE 0x2000u /* ss_lufs_norm_batch_f32 with SS_FLAG_DEVICE_PTR: `result` is a DEVICE array of 4 * S doubles; the call only enqueues work (no synchronisation) */
#define SS_FLAG_KEEP_SPEAKERS 0x4000u /* ss_mix_f32: do not write the scaled interferers back into `speakers` (read-only input; the mix is the only output) */
#define SS_FLAG_BANK_DEVICE 0x8000u /* render entry points WITHOUT SS_FLAG_DEVICE_PTR: `rirs` alone is a DEVICE pointer (a resident bank rendered for a host dry signal into a host array) */
#define SS_FLAG_ROW_SPECTRA 0x10000u /* render entry points, assembly engine: transform EVERY filter row once in a pre-pass (default: only rows cut into >= 5 tasks, i.e. trajectories of few points, and static sources; csrc/plan.h flag_long_rows) */
#define SS_FLAG_NO_ROW_SPECTRA 0x20000u /* ... never: every task transforms its row's taps itself (the only form before round 6) */
#define SS_FLAG_BACKGROUND 0x40000u /* ss_rir_bank_synth_batch_f32: the launch runs BESIDE another stream's kernels (a scene's loudness / mix while the next scene's banks are generated): at most five workgroups per CU instead of eight, so that the other stream's workgroups find wave slots (a scene -3.5 %; the launch alone +4 %) */
#define SS_FLAG_ASYNC_PLAN 0x800u /* ss_convolve_moving_f32 with device pointers: plan the explicit schedule on the device (no host synchronisation) */

int ss_version(void);
const char* ss_last_error(void);

/* Lazy: every entry point initialises the current HIP device on first use.  ss_init pins the
 * device explicitly (device < 0 = keep the current one).  Importing the Python shim never calls
 * this (SonicSet.py:154 uses the 'spawn' start method; workers re-import modules). */
int ss_init(int device);
int ss_shutdown(void);

/* ---- row V: SonicSim-SonicSet/SonicSim_moving.py:63-96  convolve_moving_receiver -------------
 * y[c,t] = (1-w[t]) * (x * rirs[idx[t],c])[t] + w[t] * (x * rirs[idx[t]+1,c])[t],  0 <= t < T
 * (causal, zero initial state, tail dropped -- the [..., :audio_len] crop at :86).
 * idx[T] int64 with 0 <= idx[t] <= P-2 (arbitrary, not necessarily monotone -- :89-94 is a pure
 * gather); w[T] float32.  Replaces oaconvolve (:86) + fancy-index gather (:89-90) + lerp (:94). */
int ss_convolve_moving_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L,
                           const int64_t* idx, const float* w, float* y, uint32_t flags, void* stream);

/* ---- row V without a host round trip ----------------------------------------------------------
 * By default ss_convolve_moving_f32 validates 0 <= idx[t] <= P-2 before it returns (the reference's fancy index at
 * SonicSim_moving.py:89-90 raises IndexError at the call), which costs one device-to-host copy and a stream synchronisation.
 * With SS_FLAG_DEVICE_PTR | SS_FLAG_ASYNC_PLAN the schedule is planned by a kernel on `stream` and the call only enqueues work
 * (assembly engine, i.e. filters longer than 8192 taps; for shorter filters the flag is ignored and the call validates as usual):
 * filter rows that do not exist then contribute nothing (an entry of -1 still takes its end filter, row 0) and the
 * condition is reported here instead.
 * ss_async_status synchronises `stream`, returns and clears the latched condition of the current device:
 * code 0 = none, 1 = interp_index out of range (where = first sample of the offending 1024-sample tile), 2 = schedule too
 * irregular for the device planner's task buffer (where = row-tasks needed; nothing was rendered: use the default path). */
int ss_async_status(int32_t* code, int64_t* where, void* stream);
/* The outcome of the LAST render ON `stream` if it planned its schedule on the device (SS_FLAG_ASYNC_PLAN honoured; otherwise
 * *out_of_range = -1: that render validated on the host, or the stream never rendered).  The words are kept per stream (in the stream's
 * workspace lane, round 6): renders enqueued on other streams cannot overwrite them and are not waited for.  Not latched, not cleared: *out_of_range == 1 with the first offending sample tile in *where, *too_irregular != 0 when the schedule did not fit the planner's task buffer (the output is then NaN).  Waits for the
 * stream that render ran on.  `ops.convolve_moving` validates with it: plan + render optimistically, one synchronisation, raise like the
 * reference's fancy index (SonicSim_moving.py:89-90) -- no bounds array travels to the host. */
int ss_plan_status_last(int32_t* out_of_range, int64_t* where, int32_t* too_irregular, void* stream);

/* ss_convolve_moving_f32 with the schedule planned on the device AND this call's verdict returned by the call itself (round 4): status[3] =
 * {out_of_range (1: an interp_index outside [0, P-2]; -1: the engine in use validated on the host before rendering, nothing to report),
 * first sample of the offending 1024-sample tile, too_irregular (the schedule did not fit the planner's task buffer: nothing valid was rendered)}.
 * The verdict is read before the device context's lock is released -- unlike ss_plan_status_last no other thread's render can slip between the render
 * and the read.  Round 6: the call returns as soon as the PLANNER has reported (it mirrors the verdict into pinned host words which the call polls,
 * ~25 us after the first launch starts), not when the render has finished: with device pointers y is ordered by the stream like every other entry
 * point's output (NaN-filled when the verdict is bad); with host pointers the call still returns after the download.  This is what
 * ops.convolve_moving(validate=True) calls. */
int ss_convolve_moving_checked_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L, const int64_t* idx,
                                   const float* w, float* y, uint32_t flags, void* stream, int64_t* status);

/* ---- how the render kernel's workgroups get their tasks (current device) --------------------------------------------------------
 * 1 (default): dynamic queues -- one queue per XCD holding that XCD's stretch of the LPT plan; every task (the first included) is taken
 * with an atomic ticket, one task ahead so that its round trip is never waited for.  A workgroup that gets onto the machine late --
 * another kernel (RCCL's send / recv during the scene gather of a multi-GPU run, a copy kernel of another stream, anything) holds its
 * compute unit -- finds its queue drained; the render degrades in proportion to the units held (tools/t_cu_steal.py, profiles/r02y).
 * 0: static lists -- workgroup b renders tasks b, b + n, b + 2n, ...  Equally fast on a GPU the render has entirely to itself and
 * with less run-to-run spread, but a workgroup that cannot be placed still owns its list: 4 of 256 compute units held = +62 % kernel
 * time.  Same output bits either way. */
int ss_set_task_queue(int dynamic);

/* ---- streams.  Every entry point is stream-ordered on the `stream` it is given and re-entrant per device.  The library's internal workspace
 * (input spectra, staged plan, queue heads, reduction scratch) exists once per STREAM: up to four streams per device are live at once, each in
 * its own workspace lane, so a caller that alternates two streams over independent renders (SonicSet.py:77-94 issues five per sample) gets render
 * i + 1's spectra launch and first tickets under the draining tail of render i's persistent launch -- the step then costs the render kernel
 * alone (0.182 -> 0.168 ms per config-2 render).  A fifth stream takes over the least recently used lane after synchronising
 * that lane's stream.  ss_workspace_lanes: {lanes, lanes in use, lane switches so far, takeovers (each one a stream synchronisation)}. */
int ss_workspace_lanes(int32_t* out, int32_t n);
/* Lifetime: a lane remembers the raw stream handle it belongs to until the library shuts down, and entry points that read state of other
 * lanes (ss_async_status) or take a lane over synchronise that handle -- a stream must therefore be RELEASED before the caller destroys it:
 * ss_stream_release(stream) synchronises the stream once, forgets it and frees the lane's workspace (input spectra, plans, row spectra: up to
 * a few hundred MB at config-5 shapes per lane).  Releasing a stream the library has never seen is not an error. */
int ss_stream_release(void* stream);

/* ---- host-pointer mode (flags without SS_FLAG_DEVICE_PTR): the path SonicSim_moving.py:122-125 really takes -- NumPy in, NumPy out.
 * The library moves the caller's arrays through a ring of pinned staging slots filled by a few host threads while the DMA engine drains
 * them (a single memcpy stream cannot keep the PCIe link busy), and for the implicit schedule on the assembly engine it cuts the bank into
 * chunks of whole trajectory positions: the rows of chunk k are rendered while chunk k + 1 is on the wire, and every stretch of the output
 * that no later chunk touches travels back at once.  Same bits as the device-pointer render.  Buffers the caller pinned itself
 * (ss_host_alloc, hipHostMalloc, hipHostRegister) are recognised and moved by DMA directly, without the staging copy.
 * ss_set_host_pipe: copy threads (0 = keep; default 4: more only contend for the memory system, profiles/r04a), bytes per upload slot (default 32 MiB, 6 of them; 4 download slots of <= 4 MiB),
 * bytes per bank chunk (default 24 MiB, at most 16 chunks), bind (-1 = keep; 2 = default: the copy threads follow the caller's pages -- bound to the
 * NUMA node the array being staged lives on (move_pages query), so the staging copy reads locally, ONE THREAD PER LAST-LEVEL CACHE of that node (sysfs
 * cache/index3/id: four threads on one CCD share its ~55 GB/s link and stage a 307 MB bank in 5.7 instead of 3.1 ms, profiles/r04al); 0 = left to the scheduler (measured bimodal:
 * 6.45 or 8.5 ms per config-2 render from run to run); 1 = bound to the CPUs next to the GPU (sysfs local_cpulist; slower when the caller's arrays
 * live on the other socket: 8.4 ms, profiles/r04d)) -- current device.
 * ss_host_path_stats: {seconds inside the last host-pointer render call, bytes up, bytes down, bank chunks, direct (pinned) transfers,
 * copy threads, [6..13] stage marks, [14] host-pointer calls that ended in an error and were drained on the way out, [15] bind, [16] last-level-cache
 * groups the copy threads are spread over, [17] NUMA node they follow (-2 never bound, -1 unbound)} of the current device.
 * THREAD-AFFINITY POLICY: only the library's OWN copy threads are ever bound (pthread_setaffinity_np on threads it created); the calling thread
 * and every other thread of the process keep their masks.  bind = 0 switches the binding off altogether (a host with its own placement policy,
 * e.g. SonicSet.py's mp.Pool workers pinned by the job scheduler): the copy threads then inherit the mask of the thread that made the first
 * host-pointer call.  An error return of a host-pointer render leaves nothing in flight (both copy streams and the render stream are drained). */
int ss_set_host_pipe(int threads, int64_t slot_bytes, int64_t chunk_bytes, int bind);
int ss_host_path_stats(double* out, int32_t n);
/* pinned host memory the DMA engines address directly (hipHostMalloc / hipHostFree): a caller that renders into such a buffer skips the
 * staging copy of host-pointer mode */
int ss_host_alloc(void** out, int64_t bytes);
int ss_host_free(void* p);

/* ---- rows I+V fused: SonicSim_moving.py:42-45 + :63-96 ---------------------------------------
 * Fast path of interpolate_moving_audio (SonicSim_moving.py:98-125).  The host keeps only the O(P)
 * half of setup_dynamic_interp (:32-39, the NumPy-RNG-coupled segment lengths n_k); the O(T)
 * expansion idx = repeat(arange(P-1), n_k), w = concat(linspace(0,1,n_k,endpoint=False)) (:42-45)
 * is implicit in the kernel epilogue (bit-exact float32 ramp).
 * seg_len[P-1] is ALWAYS a host pointer; seg_len[k] >= 0, sum == T. */
int ss_convolve_moving_seg_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L,
                               const int64_t* seg_len, float* y, uint32_t flags, void* stream);

/* ---- rows G+I+V: SonicSim_audio.py:398 deferred into SonicSim_moving.py:63-96 ----------------
 * y = render of (rirs / *divisor).  The global peak normalisation of a bank is one scalar and the render is linear in the
 * filters, so a bank that is only rendered -- never handed back -- need not be rewritten: the scalar
 * is applied to the dry signal while its spectra are formed.  Device pointers only (SS_FLAG_DEVICE_PTR); `divisor` is a
 * device float (e.g. the peak ss_rir_bank_synth_peak_f32 left).  Differs from rendering the materialised bank by float32
 * round-off only (1e-7 relative). */
int ss_convolve_moving_seg_div_f32(const float* x, int64_t T, const float* rirs, int32_t P, int32_t C, int32_t L,
                                   const int64_t* seg_len, const float* divisor, float* y, uint32_t flags, void* stream);

/* ---- a whole scene in ONE persistent launch (round 3).  SonicSet.py:61-94 renders three moving speakers (interpolate_moving_audio,
 * SonicSim_moving.py:98-125) and two static sources (convolve_fixed_receiver, :47-61) of the same length one after the other; here the
 * `nsrc` (1..8) renders share one spectra launch and one render launch (the tasks of all sources in one XCD-aware list), which saves a
 * spectra launch, two kernel boundaries and a load-balancing tail per extra source.  Source s: dry signal x[s][T], filters rirs[s]
 * [P[s]][C][L] (P[s] == 1: a static source, its one filter applied with coefficient 1; P[s] >= 2: a moving source with HOST segment
 * lengths seg_len[s][P[s]-1], sum = T, as in ss_convolve_moving_seg_f32), optional device scalar divisor[s] (deferred peak
 * normalisation as in ss_convolve_moving_seg_div_f32; the array or an entry may be NULL), output y[s][C][T].  The pointer ARRAYS are host
 * arrays of DEVICE pointers (SS_FLAG_DEVICE_PTR required); C < 65536, L > 128, T < 2^30 (the assembly engine's shapes).  Results are
 * bit-identical to the separate calls. */
int ss_convolve_scene_f32(int32_t nsrc, const float* const* x, int64_t T, const float* const* rirs, const int32_t* P, int32_t C, int32_t L,
                          const int64_t* const* seg_len, const float* const* divisor, float* const* y, uint32_t flags, void* stream);

/* ---- streaming render with persistent state (not in the reference, whose SonicSim_moving.py:98-125 renders a
 * whole source at once).  ss_stream_open fixes the bank rirs[P][C][L] (DEVICE pointer, must stay valid and unchanged until close) and the
 * schedule seg_len[P-1] (HOST, as in ss_convolve_moving_seg_f32; the total length is their sum).  ss_stream_push takes the next n dry
 * samples (DEVICE) and writes their rendered audio out[C][n] (DEVICE) -- the same samples a one-piece ss_convolve_moving_seg_f32 render
 * produces, to float32 round-off (the inverse transforms see differently grouped sums).  What persists between pushes, in HBM: the
 * partition spectra of the filter rows the trajectory is between (each row is transformed once, a segment ahead of its use), the ring of
 * input spectra of the completed 4096-sample blocks, the dry signal so far.  A push therefore costs one forward transform + two
 * multiply-accumulate sweeps over ceil(L / 4096) partitions + two inverse transforms per channel, whatever L: one kernel launch per
 * piece (a push is cut where it crosses a block of the global 4096-sample grid or a trajectory segment), no host synchronisation.
 * ss_stream_info: {samples pushed, total, pushes, pieces (launches), filter rows transformed, bytes of persistent state}. */
int ss_stream_open(void** handle, const float* rirs, int32_t P, int32_t C, int32_t L, const int64_t* seg_len, uint32_t flags, void* stream);
int ss_stream_push(void* handle, const float* chunk, int64_t n, float* out, uint32_t flags, void* stream);
int ss_stream_info(void* handle, int64_t* out, int32_t n);
int ss_stream_close(void* handle);

/* ---- row F: SonicSim_moving.py:47-61  convolve_fixed_receiver --------------------------------
 * y[c,t] = (x * h[c])[t], 0 <= t < T.   h[C][L]. Replaces scipy.signal.fftconvolve(...)[:, :T]. */
int ss_convolve_fixed_f32(const float* x, int64_t T, const float* h, int32_t C, int32_t L, float* y,
                          uint32_t flags, void* stream);

/* ---- row R: SonicSim-SonicSet/SonicSim_rir.py:668-721 render_ir, :611-666 create_custom_arrayir,
 *      :724-791 render_rir_parallel  (output contract only; the Habitat/RLR path tracer is an
 *      external closed-source library -> PARITY UNPINNED) ------------------------
 * Synthetic RIR bank, generated on the device:
 *   bank[p,c,t] = dgain[p,c]*[t==delay[p,c]] + tail_gain*exp(-6.91 t/(rt60*fs))*n_p[c,t]*[t>delay[p,c]]
 *   n_0 = g_0, n_p = rho*n_{p-1} + sqrt(1-rho^2)*g_p, g = counter-hash Box-Muller normal(seed,p,c,t)
 * delay/dgain are HOST pointers ([P][C], O(P*C) metadata).  bank[P][C][L] follows flags bit 0. */
typedef struct SsRirParams {
    int32_t P, C, L;
    float fs;          /* sample rate (SonicSim_rir.py:178: 16000)           */
    float rt60;        /* seconds                                              */
    float tail_gain;   /* diffuse tail level relative to a unit direct path    */
    float rho;         /* AR(1) correlation of the tail across adjacent positions */
    uint32_t seed;
    const int32_t* delay; /* [P][C] direct-path delay in samples (host)      */
    const float* dgain;   /* [P][C] direct-path gain (host)                  */
} SsRirParams;
int ss_rir_bank_synth_f32(const SsRirParams* prm, float* bank, uint32_t flags, void* stream);
/* The same, and max |bank| (the abs().max() of SonicSim_audio.py:398) leaves the generator with the bank: a wave reduction
 * plus one atomic max per wave inside the generating kernel instead of a second pass over the bank.  `peak` follows flags bit 0
 * (device float: no synchronisation; host float: returned with the bank). */
int ss_rir_bank_synth_peak_f32(const SsRirParams* prm, float* bank, float* peak, uint32_t flags, void* stream);

/* The banks of one scene in ONE launch (SonicSet.py:61-63 generate_rir_combination x 3, :86-91 render_ir x 2): prm[n] (HOST array of n <= 8
 * parameter records), banks[n] / peaks[n] (HOST arrays of DEVICE pointers; peaks or an entry may be NULL).  Needs SS_FLAG_DEVICE_PTR |
 * SS_FLAG_META_DEVICE; banks that share C * L (even L, 16-byte aligned) run as one grid of (workgroups x banks), anything else falls back
 * to n launches.  Same values as ss_rir_bank_synth_peak_f32 bank by bank. */
int ss_rir_bank_synth_batch_f32(int32_t n, const SsRirParams* prm, float* const* banks, float* const* peaks, uint32_t flags, void* stream);

/* ---- row G: SonicSim-SonicSet/SonicSim_audio.py:398  ir_output /= ir_output.abs().max() -------
 * In-place global peak normalisation (IEEE float32 division, bit-exact with the reference's
 * elementwise true division).  peak_out (HOST pointer, may be NULL) receives the peak; asking for
 * it synchronises the stream.  Degenerate banks behave like the reference: an all-zero bank becomes NaN (0/0), a NaN anywhere
 * makes everything NaN. */
int ss_peak_normalize_f32(float* data, int64_t n, float* peak_out, uint32_t flags, void* stream);
/* Row R, optional ("a geometric RIR model richer than K1 behind render_ir"): image-source early reflections of a
 * shoebox room [0, room] added onto bank[P][C][L] (device pointer).  src[P][3], mic[C][3] (metres, inside the box), pat[P][C] (channel
 * pattern of the direct path), room[3]: HOST arrays.  Every image with 1..order wall reflections contributes
 * pat * beta^reflections / max(d, 0.1) at the fractional delay fs * d / 343, split linearly over the two neighbouring taps (the first
 * min(L, 16384) taps).  Content is synthetic by definition (oracle/rir_synth.py::early_reflections); order 0 is a no-op. */
int ss_rir_early_add_f32(float* bank, int32_t P, int32_t C, int32_t L, float fs, const float* src, const float* mic, const float* pat,
                         const float* room, float beta, int32_t order, uint32_t flags, void* stream);

/* data /= *divisor with a divisor that is already known (ss_rir_bank_synth_peak_f32): the single pass that materialises the
 * normalised bank of SonicSim_audio.py:398 when the caller wants the bank itself (SonicSet.py:68 saves it).  `divisor`
 * follows flags bit 0.  Same IEEE division, same bits as ss_peak_normalize_f32. */
int ss_divide_by_f32(float* data, int64_t n, const float* divisor, uint32_t flags, void* stream);

/* ---- row M: separation/look2hear/datas/movingdatamodule.py:29-32 compute_mch_rms_dB ----------
 * out_db[i] = 10*log10(max(1e-20, mean(x_i^2))) over ALL n elements of each of `count` equally
 * sized arrays x[count][n].  out_db is a HOST pointer (synchronises). */
int ss_rms_db_f32(const float* x, int64_t n, int32_t count, double* out_db, uint32_t flags, void* stream);

/* ---- row M: movingdatamodule.py:105-124 (twin at :205-224) SIR/SNR mix -----------------------
 * speakers[S][n] (interferers 1..S-1 are scaled IN PLACE like :113), noises[N][n],
 * sirs[S-1] (host), snr (dB).  mix[n] = sum_s speakers[s] + g_n * sum_k noises[k] with
 *   g_i = 10^(min(E(spk0) - E(spk_i) - sir_i, 40)/20),  g_n = 10^(min(E(speech) - E(noise) - snr, 40)/20).
 * gains_out (host, may be NULL) receives [g_1..g_{S-1}, g_n].  No host synchronisation unless
 * gains_out != NULL or host-pointer mode.
 * Reproducibility: the energies behind the gains are float64 sums whose ORDER depends on whether the 16-byte-vector kernels apply (n % 4 == 0
 * and 16-byte aligned pointers) or the scalar ones; the same stems give bit-identical gains run to run and for every buffer of the same
 * alignment class, but a crop that starts at an odd offset (scalar form) may differ from the aligned form in the last bits of a gain (1e-16
 * relative in the energies).  The goldens of movingdatamodule.py pin the aligned form. */
int ss_mix_f32(float* speakers, int32_t S, const float* noises, int32_t N, int64_t n, const float* sirs,
               float snr, float* mix, float* gains_out, uint32_t flags, void* stream);

/* ---- row N2: the dataset-side step after the path -- separation/look2hear/datas/movingdatamodule.py:56-126 (MovingTrainDataset
 *      .__getitem__: mono fold, random crop, -40 dB silence rejection, SIR/SNR mix), twins in enhancement/look2hear/datas/.
 *      Rendered stems stay in HBM; the Python ``random`` / torch RNG draws stay on the host (their order IS the contract).
 *      All four take device pointers only (SS_FLAG_DEVICE_PTR). ---------------------------------------------------------------
 * out[t] = (x[0][t] + ... + x[C-1][t]) / C: ``wav.mean(dim=0)`` (:63, :77), float32 sum in channel order, IEEE division. */
int ss_mean_channels_f32(const float* x, int32_t C, int64_t T, float* out, uint32_t flags, void* stream);
/* compute_mch_rms_dB (:29-32) of K crops in one launch: crop k = C rows of n samples, row c at crops[k] + c * chan_stride
 * (crops: HOST array of K device pointers).  out_db[K] (host) = 10 log10(max(1e-20, mean over all C*n elements)), float64
 * accumulation.  Synchronises (the -40 dB rejection loop at :84-100 needs the answer before its next random draw). */
int ss_crop_rms_db_f32(const float* const* crops, int32_t K, int32_t C, int64_t chan_stride, int64_t n, double* out_db,
                       uint32_t flags, void* stream);
/* :104-124 for B dataset items in one launch sequence.  speakers[B*S], noises[B*N]: HOST arrays of device pointers to the
 * crop starts (C rows of n samples, chan_stride apart).  sirs[B*(S-1)], snrs[B] (host) are the drawn values.  Writes
 * speakers_out[B][S][C][n] (interferers 1..S-1 scaled by 10^(min(E_0 - E_i - sir, 40)/20), the copy the reference returns) and
 * mix_out[B][C][n] = sum of the scaled speakers + 10^(min(E_speech - E_noise - snr, 40)/20) * sum of the noises, all float32 in
 * the reference's order of operations; energies in float64.  gains_out[B*S] (host, may be NULL; asking synchronises). */
int ss_mix_batch_f32(const float* const* speakers, const float* const* noises, int32_t B, int32_t S, int32_t N, int32_t C,
                     int64_t chan_stride, int64_t n, const float* sirs, const float* snrs, float* speakers_out, float* mix_out,
                     float* gains_out, uint32_t flags, void* stream);
/* enhancement/look2hear/datas/movingdatamodule.py:34-48 overlap_audio: out[t] = (x[t-d] + x[t+d]) + x[t] with zero fill. */
int ss_overlap_audio_f32(const float* x, float* out, int64_t T, int64_t delay_samples, uint32_t flags, void* stream);
/* enhancement/look2hear/datas/movingdatamodule_remix.py:136-146 (the "remix" training item: speech and noise crops summed WITHOUT
 * level randomisation): out[t] = (first_0[t] + first_1[t] + ...) + (second_0[t] + ...), float32, left to right inside a group like
 * torch.sum over the stack dimension.  first / second: HOST arrays of device pointers to the crop starts; at most 8 sources in all. */
int ss_crop_sum_f32(const float* const* first, int32_t n_first, const float* const* second, int32_t n_second, int64_t n, float* out,
                    uint32_t flags, void* stream);

/* ---- row N3: the source-assembly step before the path -- torchaudio.transforms.Resample(orig_freq=sr, new_freq=sample_rate)
 *      at SonicSim-SonicSet/SonicSim_audio.py:249,297 (44.1 / 48 kHz corpora -> 16 kHz).  PARITY UNPINNED (torchaudio is absent): the
 *      published sinc_interp_hann algorithm, see oracle/resample.py.
 * x[rows][L] -> out[rows][Lout], Lout = ceil(nnew * L / orig) with orig / nnew already divided by their gcd:
 *   out[r][f * nnew + p] = sum_{j < ntap} taps[j][p] * x[r][f * orig - width + first[p] + j]   (zeros outside [0, L))
 * taps[ntap][nnew] (HOST, tap-major) are the non-zero taps of the (2 * width + orig)-tap kernel of phase p, first[nnew] (HOST) the
 * index of each phase's first kept tap -- built by the host side from the published formula (sonicsim_amd/resample.py).
 * x / out follow flags bit 0. */
int ss_resample_f32(const float* x, int32_t rows, int64_t L, int32_t orig, int32_t nnew, int32_t width, const float* taps,
                    const int32_t* first, int32_t ntap, float* out, int64_t Lout, uint32_t flags, void* stream);

/* ---- row U: SonicSim-SonicSet/SonicSim_audio.py:68-81 lufs_norm (pyloudnorm.Meter) -----------
 * BS.1770-4 K-weighted mean-square per gating block:  z[c][j] = sum_{t in [lo_j,hi_j)} k(x_c)[t]^2 / norm
 * where k() is the two-biquad K-weighting cascade (float64 state, coefficients coef[2][6] =
 * {b0,b1,b2,a0,a1,a2} per stage, HOST pointer), block bounds lo/hi [nblocks] int64 (HOST).
 * audio is [C][T] (or [T][C] with SS_FLAG_LAYOUT_TC).  z_out[C][nblocks] float64 is a HOST pointer
 * (synchronises).  The gating arithmetic (O(blocks)) stays on the host like segment lengths do. */
int ss_kweighted_block_power_f32(const float* audio, int64_t T, int32_t C, const double* coef,
                                 const int64_t* lo, const int64_t* hi, int32_t nblocks, double norm,
                                 double* z_out, uint32_t flags, void* stream);

/* out[i] = gain * in[i]  (pyloudnorm.normalize.loudness, called at SonicSim_audio.py:77);
 * sums_out (host, may be NULL): {sum(out), sum(in)} in float64 for the gain = n/d of :78-79. */
int ss_scale_f32(const float* in, float* out, int64_t n, float gain, double* sums_out, uint32_t flags,
                 void* stream);

/* lufs_norm in ONE call (SonicSim_audio.py:68-81 incl. pyloudnorm's Meter.integrated_loudness gating and
 * normalize.loudness): block powers as above, the BS.1770-4 two-stage gating and the gain on the device,
 * out = (float)gain * audio.  weights[C] (HOST) are the channel weights G; target_lufs is the drawn class
 * loudness.  result (HOST, 4 doubles): {integrated loudness (-inf if no block survives the gates; the gain
 * then uses -40 like the reference), linear gain, sum(out), sum(audio)}.  One synchronisation at the end. */
int ss_lufs_norm_f32(const float* audio, float* out, int64_t T, int32_t C, const double* coef,
                     const int64_t* lo, const int64_t* hi, int32_t nblocks, double block_norm,
                     const double* weights, double target_lufs, double* result, uint32_t flags,
                     void* stream);

/* The same for S stems in one call (one launch sequence, one synchronisation): audio/out are [S][C][T]
 * channel-first contiguous, targets[S] (HOST) the drawn class loudness of every stem, result[S][4] as above.
 * S <= 16 and S * C <= 64. */
int ss_lufs_norm_batch_f32(const float* audio, float* out, int64_t T, int32_t C, int32_t S,
                           const double* coef, const int64_t* lo, const int64_t* hi, int32_t nblocks,
                           double block_norm, const double* weights, const double* targets,
                           double* result, uint32_t flags, void* stream);

/* ---- rows U + M back to back (SonicSet.py:97-101 normalises the stems, movingdatamodule.py:105-124 mixes them): the energy the mix needs of
 * every stem rides on the pass that writes the normalised stem.
 * ss_lufs_norm_batch_sq_f32 = ss_lufs_norm_batch_f32 with SS_FLAG_DEVICE_PTR | SS_FLAG_RESULT_DEVICE (no synchronisation) that also leaves
 * sum(out[s]^2) (float64, every element of stem s) in the DEVICE array sumsq[S].
 * ss_mix_presum_f32 = ss_mix_f32 for ONE noise stem whose speaker / noise energies are already known (device doubles sumsq_speakers[S],
 * sumsq_noise[1], e.g. entries of the array above): two launches instead of five, 184 MB instead of 276 MB of traffic for a 2-speaker
 * 8 x 960 000 mix.  The same arithmetic per sample as ss_mix_f32; the energies are float64 sums in another association, so a gain may differ
 * from ss_mix_f32's in its last bit (the north-star gate is 1e-4).  Device pointers, 16-byte aligned stems, n % 4 == 0; gains_dev (optional,
 * device, S + 1 floats) receives {1, interferer gains..., noise gain}.  SS_FLAG_KEEP_SPEAKERS as in ss_mix_f32. */
int ss_lufs_norm_batch_sq_f32(const float* audio, float* out, int64_t T, int32_t C, int32_t S,
                              const double* coef, const int64_t* lo, const int64_t* hi, int32_t nblocks,
                              double block_norm, const double* weights, const double* targets,
                              double* result, double* sumsq, uint32_t flags, void* stream);
int ss_mix_presum_f32(float* speakers, int32_t S, const float* noise, int64_t n, const float* sirs, float snr, float* mix,
                      const double* sumsq_speakers, const double* sumsq_noise, float* gains_dev, uint32_t flags, void* stream);

/* ---- round 6: the mix in ONE pass.
 * ss_lufs_norm_batch_sqx_f32 = ss_lufs_norm_batch_sq_f32 whose first nspk stems (2 <= nspk <= 4) are the speakers of the mix that follows: the pass that
 * writes the normalised stems also leaves their cross sums sum(out[i] * out[j]) behind the S energies, sums[S + j (j - 1) / 2 + i] for i < j < nspk
 * (the workgroups of speaker j read the inputs of the speakers before it once more: +1 stem of reads for two speakers).  Stems 16-byte aligned, C * T % 4 == 0.
 * ss_mix_onepass_f32 = ss_mix_presum_f32 with those sums (sumsq_speakers[S] and cross_speakers[S (S - 1) / 2] in the order above, device doubles): the energy of
 * the speech sum s_0 + sum g_s s_s (movingdatamodule.py:113-118) follows from them in float64, so interferer gains, noise gain and the mix itself are ONE
 * launch reading every stem once -- 123 MB instead of 184 MB for a 2-speaker 8 x 960 000 mix.  Per sample the same float32 operations; the speech energy
 * is the exact quadratic form instead of the sum over the float32-rounded speech sum (1e-10 relative apart), so the noise gain may differ in a last bit. */
int ss_lufs_norm_batch_sqx_f32(const float* audio, float* out, int64_t T, int32_t C, int32_t S, int32_t nspk,
                               const double* coef, const int64_t* lo, const int64_t* hi, int32_t nblocks,
                               double block_norm, const double* weights, const double* targets,
                               double* result, double* sums, uint32_t flags, void* stream);
int ss_mix_onepass_f32(float* speakers, int32_t S, const float* noise, int64_t n, const float* sirs, float snr, float* mix,
                       const double* sumsq_speakers, const double* cross_speakers, const double* sumsq_noise, float* gains_dev, uint32_t flags,
                       void* stream);

/* ---- measurement hooks (bench.py): HIP-event timing of the dominant kernels on their own stream.
 * kind 0 = overlap-save render kernel (one parity pass = one launch), 1 = input-spectra kernel,
 * 2 = direct-form kernel.  ss_prof_enable(0) = off, 1 = every launch, N > 1 = every N-th launch of each
 * kind (an event pair costs two barrier packets).  ss_prof_read synchronises, then returns the number of
 * timed launches and their total milliseconds accumulated since the last ss_prof_enable(on != 0). */
int ss_prof_enable(int on);
int ss_prof_read(int kind, int64_t* launches, double* total_ms);
/* all launches of `kind` since ss_prof_enable(on != 0), timed or not */
int ss_prof_seen(int kind, int64_t* launches);
/* the individual durations (ms, launch order) of the timed launches of `kind`: at most `cap` are written, *launches = how many
 * there are (bench.py reports min / median / p90 / max of the render kernel over the timed region) */
int ss_prof_list(int kind, double* ms_out, int64_t cap, int64_t* launches);

/* ---- multi-GPU gather WITHOUT compute units ("ss_gather_scenes"; round 6).  SonicSet.py:183-211 renders its scenes in
 * one serial loop; here scenes shard over one process per GPU and only the finished (C, T) stems travel to the root.  The default path
 * (sonicsim_amd/parallel.py SceneGather) moves them with RCCL send / recv, whose kernels take compute units from a render kernel built
 * around owning all 256.  This path does not: the root exports its result array `scenes[num_scenes][scene_bytes]` as a HIP IPC handle,
 * every other rank opens it and copies each finished scene straight into its slot with the copy engines (SDMA over xGMI / the local
 * fabric: hipMemcpyAsync device-to-device on a copy stream of its own, ordered behind the render by an event).
 *   root:   ss_gather_create(&g, num_scenes, scene_bytes, ipc_out)   -- allocates the array; the 64 bytes of ipc_out go to the other
 *                                                                      ranks through the host framework's control plane (any transport)
 *   others: ss_gather_attach(&g, ipc_in, num_scenes, scene_bytes)
 *   all:    ss_gather_slot(g, i, &ptr)     root only: where scene i lives (render in place: nothing to copy)
 *           ss_gather_put(g, i, src, stream)   enqueue "copy scene i from device memory `src` once `stream` has reached this point"; returns at
 *                                              once; `src` may be rewritten after ss_gather_wait_src(g, stream) / ss_gather_flush
 *           ss_gather_flush(g)             every put of this process has landed in the root's memory (host-side wait)
 *           ss_gather_close(g)
 * After all ranks flushed and met at a host barrier the root may read ss_gather_slot(g, i).  Device pointers only. */
#define SS_IPC_HANDLE_BYTES 64
int ss_gather_create(void** handle, int64_t num_scenes, int64_t scene_bytes, void* ipc_handle_out /* SS_IPC_HANDLE_BYTES */);
int ss_gather_attach(void** handle, const void* ipc_handle_in, int64_t num_scenes, int64_t scene_bytes);
int ss_gather_slot(void* handle, int64_t scene, void** ptr);
int ss_gather_put(void* handle, int64_t scene, const void* src, void* stream);
int ss_gather_wait_src(void* handle, void* stream);      /* `stream` waits for every put enqueued so far (their sources may then be rewritten by work on `stream`) */
int ss_gather_flush(void* handle);
int ss_gather_close(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* SONICSIM_HIP_H */
