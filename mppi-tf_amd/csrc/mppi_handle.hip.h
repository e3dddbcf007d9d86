// mppi_handle.hip.h — host-side state of one controller (mppi_handle) and the launchers of the rollout kernels.
// The library is several translation units compiled in parallel (mppi-tf_amd/build.py): mppi_capi.hip holds the C-ABI,
// mppi_launch_tile.hip / _pc.hip / _mlp.hip instantiate one kernel family each for ONE action dimension per object
// (-DMPPI_UNIT_A=1..4), mppi_launch_gen.hip the generic-model kernels. This header is what they share.
#pragma once
#include "mppi_kernels.hip.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <string>
#include <vector>

using namespace mppi;

// ----------------------------------------------------------------------------------------
struct mppi_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    DevConsts hc{};
    DevConsts *dC = nullptr;
    int norm_two_pass = 0;    // this step's records come from the second pass of the two-pass normalizeCost path (set per step)
    int K_global = 0, K_local = 0, k_offset = 0, shard_rank = 0, shard_count = 1;
    int H = 0, s = 0, a = 0, HA = 0;
    int R = 64, nb = 0;   // tile size / record count of the point-mass tile kernels
    int nb_mlp = 0;       // record count of the MLP rollout kernel (64 rollouts per workgroup)
    int nbp = 0;          // record slots in d_part: record_pad(max(nb, nb_mlp)), the column stride of every rollout launch
    int part_nb = 0;      // tile count whose slots currently hold records (0: all slots neutral)
    int fp_contract = 0;  // MPPI_FLAG_FP_CONTRACT: the point-mass producer/consumer rollout with fused multiply-adds (PC_COST_DIAG_FMA)
    int mlp_bx3 = 0;      // MPPI_FLAG_MLP_BF16X3: split-bf16 matrix-core variant of the MLP rollout
    int mlp_small = 0;    // hidden width (16 or 32) of a small learned model served by k_rollout_mlp_small, else 0
    MlpSmallArgs small_args{};
    int mlp32_valu = 0;   // tuning: a Dense(32) network on 1 = the vector-ALU kernel, 2 = the one-wave-per-32-rollouts matrix-core kernel, instead of the two-wave pipeline
    int n_cu = 256;       // compute units of the device (k_rollout_mlp2 runs one tile-walking workgroup per CU)
    int mlp_v2 = 0;       // exact-fp32 MLP rollouts run k_rollout_mlp2 (one wave per SIMD, two pipelined sets; a_dim <= 3)
    MlpDev hm{};          // learned model: device pointers + normalisation (host copy)
    MlpDev *dM = nullptr;
    float *d_mlp_w = nullptr; // one allocation holding W1,b1,W2,b2,W3,b3
    size_t tile_lds = 0;
    int normalize = 0;
    int sigma_diag = 0; // Σ and Σ⁻¹ are exactly diagonal (the DIAG kernel instances are bit-identical then)
    int pc_np = 5;      // producer waves per workgroup of k_rollout_pc (chosen by tiles per CU; MPPI_TUNE_PC_PRODUCERS overrides)
    int pc_pass = 0;    // which pass of k_rollout_pc the next launch_pc is: 0 the step's one pass, 1 / 2 the two passes of normalizeCost (set per launch by mppi_capi.hip)
    int pc_range_given = 0; // the weights-only pass takes its temperature from d_mm[2] (a K-sharded handle: the ranks agreed on the range) instead of reducing d_tile_mm
    float *d_tile_mm = nullptr; // normalize_cost: every tile's (min, max) cost of the first pass, [2][nbp]
    // diagnostic switches, set only through mppi_set_tuning (the library reads no environment variable)
    int force_tile = 0;   // MPPI_TUNE_FORCE_TILE_KERNEL: the LDS-tile kernel instead of the producer/consumer one (A/B timing)
    int pc_no_balance = 0; // MPPI_TUNE_PC_BALANCE = 0: no SIMD-true roles / progress priorities
    int pc_bias = -1;      // head start (quarter chunks) of the four generations of workgroups on a CU, gen 0 in the low nibble (pc_set_prio); -1: the launcher's choice; MPPI_TUNE_PC_BALANCE = 0x10000 | biases
    int pc_lds_min = 0;   // MPPI_TUNE_PC_LDS_MIN: pad the dynamic LDS (caps workgroups per CU)
    int sync_spin = 1;    // MPPI_TUNE_SYNC_SPIN: the synchronous step watches the pinned u slot (0: waits for the stream)
    int p2p_fault = 0;    // MPPI_TUNE_P2P_FAULT: 1 = inbox export refused, 2 = probe reports failure (fallback tests)
    int trace = 0;        // MPPI_TUNE_TRACE: roctx ranges around what a step enqueues
    int gen_one_wave = 0; // MPPI_TUNE_GEN_ONE_WAVE: the Fossen AUVModel on k_rollout_gen<0> (one wave per tile) instead of k_rollout_auv_pc
    float *d_x = nullptr, *d_u = nullptr, *d_cost = nullptr, *d_cost2 = nullptr;
    // The nominal sequence lives in one of two buffers of tau*a + a floats whose last a floats stay zero. A step
    // reads U from ubuf[u_cur] + u_off and writes U' to the other buffer at offset 0; the shifted sequence
    // (mShift + mInit0, controller_base.cpp:310-324) is then simply that buffer read from offset a_dim.
    float *d_Ubuf[2] = {nullptr, nullptr};
    int u_cur = 0, u_off = 0;
    float *U_cur() const { return d_Ubuf[u_cur] + u_off; }
    float *U_other() const { return d_Ubuf[1 - u_cur]; }
    float *d_Uupd = nullptr; // U' of the last step (MPPI_DBG_U_UPDATED)
    void U_advance() { u_cur = 1 - u_cur; u_off = a; d_Uupd = d_Ubuf[u_cur]; }
    // options of the Python reference's update: clip_act limits [a_min | a_max] and the Savitzky-Golay filter
    float *d_clip = nullptr;
    int sg_window = 0;
    float *d_sg_rows = nullptr;
    int *d_sg_start = nullptr;
    float *d_part = nullptr, *d_part2 = nullptr, *d_part3 = nullptr, *d_record = nullptr, *d_dbg = nullptr, *d_mm = nullptr;
    float *d_eps = nullptr; // lazily allocated [K_local, H, a] for injected noise / debug export
    float *d_recs = nullptr, *d_range = nullptr; // mppi_shard_step: all shards' records (the own one in place) and the {-min, max} pair, lazily
    unsigned long long *d_step = nullptr;
    // pinned, device-mapped host staging for the synchronous path: x slot 0 | x slot 1 | u. The kernels read
    // x and write u straight through these (zero-copy over PCIe, 24 B / 12 B): no H2D / D2H copy nodes per step.
    float *h_pin = nullptr, *d_pin = nullptr;
    int pin_slot = 0;
    std::string err;
    // profiling: event pairs around the rollout / finish kernels (mppi_profile_begin/end)
    std::vector<hipEvent_t> ev;   // 4 events per step: rollout begin/end, finish begin/end
    int prof_cap = 0, prof_n = 0; // steps that can be / have been recorded
    hipStream_t prof_stream = nullptr;
    // when a step is being profiled the dominant kernel is launched with hipExtLaunchKernel, whose start/stop events
    // carry the dispatch's own begin/end timestamps (what rocprofv3 reports), not the stream-level gaps around it
    hipEvent_t kev0 = nullptr, kev1 = nullptr;
    std::string no_rollout; // non-empty: why this handle cannot run rollouts (helpers still work)
    int is_gen = 0;         // model_base is AUVModel / NNAUVModel: rollouts run k_rollout_gen
    void *gen = nullptr;    // 13-state AUV family (model AUV / NN_AUV; mppi_launch_gen.hip owns it): constants + device copy
    // transition log (m_db of the reference: addX/addU/addNext/toCSV, data_base.cpp:29-71): off until
    // mppi_set_transition_log gives it a capacity; a ring of rows (x | u | x_next | has_next), allocated once there
    std::vector<float> log_rows;
    size_t log_cap = 0, log_count = 0, log_head = 0; // capacity in rows, rows held, index of the oldest row
    unsigned long long log_dropped = 0;               // rows overwritten since the log was (re)sized: the ring was full
    size_t log_stride() const { return (size_t)2 * s + a + 1; }
    // direct record exchange (mppi_shard_p2p_*): own inbox, the peers' mapped inboxes, call sequence number
    unsigned long long *xchg_inbox = nullptr;
    XchgPeers xchg_peers{};
    std::vector<void *> xchg_opened; // hipIpcOpenMemHandle mappings to close
    bool xchg_attached = false;
    unsigned xchg_seq = 0, probe_seq = 0;
    long long xchg_timeout_ticks = 0;
    unsigned *h_xchg_status = nullptr, *d_xchg_status = nullptr; // pinned, device-mapped: deadline flag for the host
    unsigned *d_xchg_dead = nullptr;                             // the same flag in device memory, read by every launch
    float *d_probe_got = nullptr;
    // r05: the whole step in one launch / the armed launch (mppi_step.hip.h, mppi_launch_step.hip)
    int fuse_step = 1;          // MPPI_TUNE_FUSED_STEP: 1 = a handle of <= 128 tiles runs its device-resident step as ONE launch (0: rollout + finish; 2: ONE launch, five producer waves at every horizon)
    int arm_us = 0;             // MPPI_TUNE_ARMED_US: soft deadline of an armed launch in microseconds, 0 = mppi_next never arms
    int arm_always = 0;         // MPPI_TUNE_ARMED_ALWAYS: arm after every mppi_next, whatever the gap between the last two calls was
    unsigned long long *d_step_recs = nullptr; // record granules of the fused step [(2 + HA)][128]
    unsigned long long *d_xslot = nullptr;     // x granules, stored by the HOST straight into fine-grained device memory (large BAR); NULL: no armed launches
    unsigned long long *d_decision = nullptr;  // tile 0's verdict on an armed launch (device)
    unsigned long long *h_arm = nullptr, *d_arm = nullptr; // pinned, device-mapped: [0] the verdict for the host, [1] the sticky error word
    unsigned step_seq = 0;      // launch sequence numbers of fused / armed launches: 31 bits, never 0, never reused
    bool arm_inflight = false;  // an armed launch of sequence number arm_seq sits in the stream, waiting for x
    unsigned arm_seq = 0;
    long long last_next_ns = 0; // steady-clock time of the previous mppi_next (the arming rule looks at the gap)
    // the pre-launched pipelined step (MPPI_TUNE_PRELAUNCH; k_step_pc<.., STEP_PRE>): steps alternate between `stream` and `stream2`
    int prelaunch = 0;            // tuning: mppi_next_device on the handle's own stream pre-launches (0: off)
    hipStream_t stream2 = nullptr;
    hipEvent_t pre_ev = nullptr, pre_ev2 = nullptr; // start-up order (pre_step, mppi_capi.hip)
    unsigned long long *d_ugr = nullptr; // [2][HA] sequence granules {value, tag}
    int *d_cu_ctr = nullptr;             // [4096] workgroup arrivals per CU (role placement of a pre-launched grid)
    bool pre_active = false;      // steps are in flight on both streams; the plain U buffers / step counter are valid again after pre_quiesce
    int pre_buf = 0;              // which half of d_ugr holds the sequence the NEXT launch reads
    unsigned pre_tag = 0;         // ... and its tag
    unsigned long long pre_step = 0; // host mirror of the Philox step counter
    unsigned long long pre_count = 0; // launches since the mode was entered (parity = stream)
    unsigned next_seq() { step_seq = (step_seq + 1u) & 0x7fffffffu; if (step_seq == 0u) step_seq = 1u; return step_seq; }
    size_t xchg_step_slots() const { return (size_t)2 * HA * shard_count * 3; }
    size_t xchg_inbox_bytes() const { return sizeof(unsigned long long) * (xchg_step_slots() + (size_t)2 * shard_count); }
};


// The dynamic-LDS ceiling (hipFuncAttributeMaxDynamicSharedMemorySize) belongs to the (kernel instance, device) pair,
// not to a handle: several handles, devices and host threads share one template instance. It is kept process-wide and
// only ever RAISED (mppi_capi.hip), so a handle that needs less never lowers it under one that needs more.
hipError_t mppi_raise_lds_ceiling(const void *kernel, int device, size_t bytes);

// launchers: one definition per action dimension, each in its own object file
#define MPPI_TILE_PARAMS mppi_handle *h, hipStream_t st, int src, int mode, const float *x_dev, const float *U_dev, \
                         const float *eps, float *cost, float *part, float *noise_out
#define MPPI_MLP_PARAMS mppi_handle *h, hipStream_t st, int src, int mode, const float *x_dev, const float *U_dev, \
                        const float *eps, float *cost
#define MPPI_PC_PARAMS mppi_handle *h, hipStream_t st, const float *x_dev
#define MPPI_DECL_A(NAME, PARAMS) \
    hipError_t NAME##1(PARAMS); hipError_t NAME##2(PARAMS); hipError_t NAME##3(PARAMS); hipError_t NAME##4(PARAMS);
MPPI_DECL_A(mppi_launch_tile_a, MPPI_TILE_PARAMS)
MPPI_DECL_A(mppi_launch_pc_a, MPPI_PC_PARAMS)
MPPI_DECL_A(mppi_launch_mlp_a, MPPI_MLP_PARAMS)
// one launch of k_step_pc (mppi_step.hip.h): mode = STEP_FUSE | STEP_ARM bits; the sequence it reads / writes is given explicitly
// (an armed launch for step n+1 is enqueued before step n's bookkeeping is committed)
struct mppi_step_launch {
    int mode; const float *x_dev, *U_in; float *U_out, *u_out; unsigned seq;
    const unsigned long long *ugr = nullptr; unsigned utag = 0; unsigned long long step_index = 0; // STEP_PRE: this step's sequence as granules, their tag, the Philox step index
    unsigned long long *ugr_out = nullptr; // STEP_PRE | STEP_FUSE: the next step's granules (the column waves write them)
};
#define MPPI_STEP_PARAMS mppi_handle *h, hipStream_t st, const mppi_step_launch *L
MPPI_DECL_A(mppi_launch_step_a, MPPI_STEP_PARAMS)
#undef MPPI_DECL_A
// the 13-state AUV family (mppi_launch_gen.hip)
const char *mppi_gen_fill(mppi_handle *h, const mppi_config *cfg); // NULL = ok, else why the config is invalid
hipError_t mppi_gen_upload(mppi_handle *h);
void mppi_gen_destroy(mppi_handle *h);
hipError_t mppi_launch_gen(mppi_handle *h, hipStream_t st, int src, int mode, const float *x_dev, const float *U_dev, const float *eps,
                           float *cost, float *part, float *noise_out);
const char *mppi_gen_kernel_name(const mppi_handle *h);
hipError_t mppi_gen_model_step(mppi_handle *h, hipStream_t st, const float *x, int kx, const float *v, int k, float *scratch, float *out_next);
hipError_t mppi_gen_costs(mppi_handle *h, hipStream_t st, const float *x, const float *u, const float *eps, int k, float *os, float *oa, float *ot);
hipError_t mppi_gen_auv_pieces(mppi_handle *h, hipStream_t st, const float *x, const float *u, int k, float *out);
hipError_t mppi_gen_e3_terms(mppi_handle *h, hipStream_t st, const float *x, int k, int in_plane, float *out);
#define MPPI_CAT_(a, b) a##b
#define MPPI_CAT(a, b) MPPI_CAT_(a, b)
