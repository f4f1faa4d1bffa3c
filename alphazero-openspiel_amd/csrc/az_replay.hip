// az_replay.hip — device-resident replay store: FIFO of games, Trainer.remove_duplicates, batch gather.
// (reference: train.py:107-120 sampling, 156-201 remove_duplicates, 226-236 FIFO.)
//
// Storage = one ring of examples (SoA, capacity `cap`), logical index i -> physical (head + i) % cap, in the
// order the reference's flattened buffer has: games oldest -> newest, plies in order:
//   key u64 (hash of the action history) | bb0, bb1 u64 | ply i32 | z f64 | pi f64[A] (dense)
// and a host-side ring of game lengths for FIFO eviction.  HBM-bound integer/byte work; no MFMA anywhere.
//
// remove_duplicates, exactly: stable radix sort of (key, logical index) -> equal keys are adjacent and in buffer
// order -> one wave per segment, lane = action, each lane adds its pi component over the members IN ORDER in
// float64 (one rounding per addition, as `[sum(x) for x in zip(acc, item)]` does), divides by the count and
// writes the average back into the first member (the reference's aliasing side effect, train.py:191-197)
// -> unique list = first members in ascending buffer order (dict insertion order).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <string.h>

#include <deque>
#include <string>
#include <vector>

#include "../../include/az_replay.h"
#include "az_engine_internal.h"

struct az_replay {
    az_replay_config cfg;
    std::string err;
    AzGeom geom;
    int A = 0, maxc = 0, max_plies = 0;
    PwPlan pw;
    int64_t cap = 0, head = 0, n = 0; // example ring
    std::deque<int32_t> game_len;     // FIFO of games (lengths), oldest first
    int64_t capacity_games = 0, dropped = 0, n_unique = 0;
    uint64_t sample_calls = 0;
    // device
    uint64_t *key = nullptr, *key2 = nullptr, *bb0 = nullptr, *bb1 = nullptr; // key2: an independent second hash of the history
    unsigned int *faults = nullptr;                                             // AZ_REPLAY_FAULT_* bits
    int32_t *ply = nullptr;
    double *z = nullptr, *pi = nullptr;
    int64_t *unique = nullptr; // [n_unique] logical indices of the first occurrences, ascending
    // staging for append
    void *stage = nullptr;
    size_t stage_bytes = 0;
};
static std::string g_replay_err;

#define RCHK(r, call)                                                     \
    do {                                                                  \
        hipError_t _s = (call);                                           \
        if (_s != hipSuccess) {                                           \
            (r)->err = std::string(#call) + ": " + hipGetErrorString(_s); \
            return AZ_E_HIP;                                              \
        }                                                                 \
    } while (0)

extern "C" const char *az_replay_last_error(const az_replay *r) { return r ? r->err.c_str() : g_replay_err.c_str(); }

extern "C" int az_replay_destroy(az_replay *r) {
    if (!r) return AZ_OK;
    (void)hipSetDevice(r->cfg.device);
    (void)hipFree(r->key);
    (void)hipFree(r->key2);
    (void)hipFree(r->faults);
    (void)hipFree(r->bb0);
    (void)hipFree(r->bb1);
    (void)hipFree(r->ply);
    (void)hipFree(r->z);
    (void)hipFree(r->pi);
    (void)hipFree(r->unique);
    (void)hipFree(r->stage);
    delete r;
    return AZ_OK;
}

static void pw_build_r(PwPlan &pw, int lo, int n) { // numpy pairwise_sum recursion (PW_BLOCKSIZE 128)
    if (n <= 128) {
        pw.lo[pw.n_blocks] = lo;
        pw.len[pw.n_blocks] = n;
        pw.ops[pw.n_ops++] = pw.n_blocks++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    pw_build_r(pw, lo, n2);
    pw_build_r(pw, lo + n2, n - n2);
    pw.ops[pw.n_ops++] = -1;
}

extern "C" int az_replay_create(const az_replay_config *cfg, az_replay **out) {
    if (!cfg || !out) {
        g_replay_err = "null argument";
        return AZ_E_INVALID;
    }
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(az_replay_config)) {
        g_replay_err = "az_replay_config.struct_size mismatch";
        return AZ_E_INVALID;
    }
    az_replay_config c = *cfg;
    if (c.game == AZ_GAME_CONNECT_FOUR) {
        c.rows = 6;
        c.cols = 7;
    } else if (c.game != AZ_GAME_BREAKTHROUGH || c.rows < 4 || c.cols < 2 || c.rows * c.cols > 64 || 6 * c.cols > 64) {
        g_replay_err = "unsupported game / board";
        return AZ_E_INVALID;
    }
    if (c.max_games < 1 || c.max_examples < 1) {
        g_replay_err = "max_games and max_examples must be >= 1";
        return AZ_E_INVALID;
    }
    az_replay *r = new az_replay();
    r->cfg = c;
    r->geom = az_make_geom(c.game, c.rows, c.cols);
    r->A = az_num_actions(c.game, c.rows, c.cols);
    r->maxc = az_max_children(c.game, c.rows, c.cols);
    r->max_plies = az_max_plies(c.game, c.rows, c.cols);
    memset(&r->pw, 0, sizeof r->pw);
    pw_build_r(r->pw, 0, r->A);
    r->cap = c.max_examples;
    r->capacity_games = c.max_games;
    if (hipSetDevice(c.device) != hipSuccess) {
        g_replay_err = "hipSetDevice failed";
        delete r;
        return AZ_E_HIP;
    }
    size_t n = (size_t)r->cap;
    bool ok = hipMalloc((void **)&r->key, n * 8) == hipSuccess && hipMalloc((void **)&r->key2, n * 8) == hipSuccess &&
              hipMalloc((void **)&r->faults, 4) == hipSuccess && hipMemset(r->faults, 0, 4) == hipSuccess &&
              hipMalloc((void **)&r->bb0, n * 8) == hipSuccess &&
              hipMalloc((void **)&r->bb1, n * 8) == hipSuccess && hipMalloc((void **)&r->ply, n * 4) == hipSuccess &&
              hipMalloc((void **)&r->z, n * 8) == hipSuccess && hipMalloc((void **)&r->pi, n * 8 * (size_t)r->A) == hipSuccess &&
              hipMalloc((void **)&r->unique, n * 8) == hipSuccess;
    if (!ok) {
        g_replay_err = "hipMalloc of the replay store failed";
        az_replay_destroy(r);
        return AZ_E_NOMEM;
    }
    *out = r;
    return AZ_OK;
}

extern "C" int az_replay_set_capacity(az_replay *r, int64_t n_games) {
    if (!r || n_games < 1 || n_games > r->cfg.max_games) return AZ_E_INVALID;
    r->capacity_games = n_games;
    return AZ_OK;
}

extern "C" int az_replay_stats_get(az_replay *r, az_replay_stats *out) {
    if (!r || !out) return AZ_E_INVALID;
    out->n_games = (int64_t)r->game_len.size();
    out->n_examples = r->n;
    out->n_unique = r->n_unique;
    out->games_dropped = r->dropped;
    out->fault_flags = 0;
    unsigned int f = 0;
    RCHK(r, hipSetDevice(r->cfg.device));
    RCHK(r, hipDeviceSynchronize());
    RCHK(r, hipMemcpy(&f, r->faults, 4, hipMemcpyDeviceToHost));
    out->fault_flags = f;
    if (f) RCHK(r, hipMemset(r->faults, 0, 4)); // reported once: the store is usable again after the caller has dealt with it
    if (f) {
        r->err = "device fault flags set:";
        if (f & AZ_REPLAY_FAULT_KEY_COLLISION) r->err += " KEY_COLLISION";
        if (f & AZ_REPLAY_FAULT_BAD_INDEX) r->err += " BAD_INDEX";
        return AZ_E_DEVICE;
    }
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t h, uint64_t v) { // splitmix64 step over (h, v)
    uint64_t z = h + 0x9E3779B97F4A7C15ull * (v + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct AppendArgs {
    // source (one generation of games, layout of az_example_view / the engine's record store), device pointers
    const int *game_len;
    const float *game_ret0;
    const uint64_t *states;
    const uint16_t *move, *child_action;
    const uint8_t *nchild;
    const uint32_t *child_visits;
    const double *value;
    const long long *dst_first; // [n_src_games] logical index of the game's first example, -1 = skip
    int n_src_games, max_plies, maxc, start_ply, A, on_policy;
    long long cap, head;
    PwPlan pw;
    uint64_t *key, *key2, *bb0, *bb1;
    int32_t *ply;
    double *z, *pi;
};

// np.sum (pairwise) of the dense length-A vector with non-zeros v[k] at act[k] (ascending), one thread.
__device__ double np_sum_sparse_serial(const PwPlan &pw, const double *v, const int *act, int nc) {
    double stack[6];
    int sp = 0, k = 0;
    for (int o = 0; o < pw.n_ops; o++) {
        int op = pw.ops[o];
        if (op < 0) {
            double b = stack[--sp], a = stack[--sp];
            stack[sp++] = a + b;
            continue;
        }
        int lo = pw.lo[op], n = pw.len[op];
        double res = 0.0;
        if (n < 8) {
            while (k < nc && act[k] < lo + n) res += v[k++];
        } else {
            double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int body = n - (n % 8), k0 = k;
            while (k < nc && act[k] < lo + n) {
                int j = act[k] - lo;
                if (j < body) r[j & 7] += v[k];
                k++;
            }
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (int kk = k0; kk < k; kk++)
                if (act[kk] - lo >= body) res += v[kk];
        }
        stack[sp++] = res;
    }
    return stack[0];
}

// one thread per (game, ply) example: state, z, dense pi from the recorded root visits
__global__ void replay_append_kernel(AppendArgs a) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int g = (int)(t / a.max_plies), i = (int)(t % a.max_plies);
    if (g >= a.n_src_games) return;
    long long first = a.dst_first[g];
    int len = a.game_len[g];
    if (first < 0 || i >= len) return;
    long long dst = (a.head + first + i) % a.cap;
    size_t src = (size_t)g * a.max_plies + a.start_ply + i;
    a.bb0[dst] = a.states[src * 2];
    a.bb1[dst] = a.states[src * 2 + 1];
    a.ply[dst] = a.start_ply + i;
    double zz = a.value[src];
    if (a.on_policy) { // game_utils.py:200-204: z_i = returns()[0] * (-1)^i
        zz = (double)a.game_ret0[g];
        if ((a.start_ply + i) & 1) zz = -zz;
    }
    a.z[dst] = zz;
    // pi: float(visit)/sum(visits) -> remove_illegal_actions (np.sum pairwise, divide) (mcts.py:161-162, alphazerobot.py:13-14)
    int nc = a.nchild[src];
    double nv[64];
    int act[64];
    long long tot = 0;
    for (int k = 0; k < nc; k++) tot += a.child_visits[src * a.maxc + k];
    for (int k = 0; k < nc; k++) {
        nv[k] = (double)a.child_visits[src * a.maxc + k] / (double)tot;
        act[k] = a.child_action[src * a.maxc + k];
    }
    double s = np_sum_sparse_serial(a.pw, nv, act, nc);
    double *out = a.pi + (size_t)dst * a.A;
    for (int x = 0; x < a.A; x++) out[x] = 0.0;
    for (int k = 0; k < nc; k++) out[act[k]] = s > 1e-6 ? nv[k] / s : 1.0 / (double)nc;
}

// one thread per game: key chain over its moves (key of ply i = hash of the first i actions)
__global__ void replay_keys_kernel(AppendArgs a) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_src_games) return;
    long long first = a.dst_first[g];
    if (first < 0) return;
    int len = a.game_len[g];
    size_t s0 = (size_t)g * a.max_plies + a.start_ply;
    uint64_t h = mix64(0x243F6A8885A308D3ull, (uint64_t)a.start_ply);
    h = mix64(h, a.states[s0 * 2]);
    h = mix64(h, a.states[s0 * 2 + 1]); // start position (identical for all games of a run)
    // second, independent chain (other seed, other per-step tweak): the reference keys on the EXACT information-state
    // string (train.py:177); (key, key2) together are a 128-bit fingerprint of the history, and the segment pass
    // raises AZ_REPLAY_FAULT_KEY_COLLISION if two members of one `key` segment differ in key2, ply or position
    uint64_t h2 = mix64(0x13198A2E03707344ull ^ (uint64_t)a.start_ply, a.states[s0 * 2] + 0x9E3779B97F4A7C15ull * a.states[s0 * 2 + 1]);
    for (int i = 0; i < len; i++) {
        a.key[(a.head + first + i) % a.cap] = h;
        a.key2[(a.head + first + i) % a.cap] = h2;
        h = mix64(h, (uint64_t)a.move[s0 + i]);
        h2 = mix64(h2 ^ 0xA4093822299F31D0ull, ((uint64_t)a.move[s0 + i] << 20) | (uint64_t)(i + 1));
    }
}

static int append_common(az_replay *r, AppendArgs &a, const std::vector<int32_t> &lens, hipStream_t st) {
    // FIFO bookkeeping on the host (game granularity), like `self.buffer.append(game)` + the trim loop
    std::vector<long long> first(lens.size(), -1);
    long long add = 0;
    for (size_t g = 0; g < lens.size(); g++)
        if (lens[g] > 0) {
            first[g] = r->n + add;
            add += lens[g];
        }
    if (add > r->cap) {
        r->err = "one generation holds more examples than max_examples";
        return AZ_E_INVALID;
    }
    // make room in the example ring first (evict oldest games if the ring would overflow)
    while (r->n + add > r->cap && !r->game_len.empty()) {
        r->head = (r->head + r->game_len.front()) % r->cap;
        r->n -= r->game_len.front();
        for (auto &f : first)
            if (f >= 0) f -= r->game_len.front();
        r->game_len.pop_front();
        r->dropped++;
    }
    long long *d_first = nullptr;
    RCHK(r, hipMalloc((void **)&d_first, sizeof(long long) * first.size()));
    RCHK(r, hipMemcpyAsync(d_first, first.data(), sizeof(long long) * first.size(), hipMemcpyHostToDevice, st));
    a.dst_first = d_first;
    a.cap = r->cap;
    a.head = r->head;
    a.A = r->A;
    a.pw = r->pw;
    a.key = r->key;
    a.key2 = r->key2;
    a.bb0 = r->bb0;
    a.bb1 = r->bb1;
    a.ply = r->ply;
    a.z = r->z;
    a.pi = r->pi;
    long long threads = (long long)a.n_src_games * a.max_plies;
    hipLaunchKernelGGL(replay_append_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(replay_keys_kernel, dim3((a.n_src_games + 255) / 256), dim3(256), 0, st, a);
    RCHK(r, hipGetLastError());
    RCHK(r, hipStreamSynchronize(st));
    (void)hipFree(d_first);
    for (size_t g = 0; g < lens.size(); g++)
        if (lens[g] > 0) r->game_len.push_back(lens[g]);
    r->n += add;
    // `while len(self.buffer) > self.n_games_buffer: del self.buffer[0]` (train.py:233-236)
    while ((int64_t)r->game_len.size() > r->capacity_games) {
        r->head = (r->head + r->game_len.front()) % r->cap;
        r->n -= r->game_len.front();
        r->game_len.pop_front();
        r->dropped++;
    }
    r->n_unique = 0;
    return AZ_OK;
}

extern "C" int az_replay_append_engine(az_replay *r, az_engine *e, void *stream) {
    if (!r || !e) return AZ_E_INVALID;
    if (e->cfg.game != r->cfg.game || e->cfg.rows != r->cfg.rows || e->cfg.cols != r->cfg.cols || e->cfg.device != r->cfg.device) {
        r->err = "engine and replay store were created for different games / devices";
        return AZ_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    RCHK(r, hipSetDevice(r->cfg.device));
    RCHK(r, hipStreamSynchronize(st));
    std::vector<int32_t> lens((size_t)e->n_games);
    RCHK(r, hipMemcpy(lens.data(), e->p.rec_len, lens.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    AppendArgs a;
    memset(&a, 0, sizeof a);
    a.game_len = e->p.rec_len;
    a.game_ret0 = e->p.rec_ret0;
    a.states = e->p.rec_states;
    a.move = e->p.rec_move;
    a.child_action = e->p.rec_child_action;
    a.nchild = e->p.rec_nchild;
    a.child_visits = e->p.rec_child_visits;
    a.value = e->p.rec_value;
    a.n_src_games = (int)e->n_games;
    a.max_plies = e->p.max_plies;
    a.maxc = e->p.maxc;
    a.start_ply = e->p.start.ply;
    a.on_policy = e->cfg.backup == AZ_BACKUP_ON_POLICY;
    return append_common(r, a, lens, st);
}

extern "C" int az_replay_append_host(az_replay *r, const az_example_view *v, int32_t start_ply, void *stream) {
    if (!r || !v || v->n_games < 1 || v->max_children != r->maxc || v->max_plies > r->max_plies || start_ply < 0) {
        if (r) r->err = "bad example view (max_children / max_plies must match the game)";
        return AZ_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    RCHK(r, hipSetDevice(r->cfg.device));
    size_t ng = (size_t)v->n_games, mp = (size_t)v->max_plies, mc = (size_t)v->max_children;
    size_t sizes[8] = {ng * 4, ng * 4, ng * mp * 16, ng * mp * 2, ng * mp * mc * 2, ng * mp, ng * mp * mc * 4, ng * mp * 8};
    const void *srcs[8] = {v->game_len, v->game_ret0, v->states, v->move, v->child_action, v->n_children, v->child_visits, v->value};
    size_t off[9] = {0};
    for (int i = 0; i < 8; i++) off[i + 1] = off[i] + ((sizes[i] + 15) & ~(size_t)15);
    if (off[8] > r->stage_bytes) {
        (void)hipFree(r->stage);
        r->stage = nullptr;
        RCHK(r, hipMalloc(&r->stage, off[8]));
        r->stage_bytes = off[8];
    }
    for (int i = 0; i < 8; i++) RCHK(r, hipMemcpyAsync((char *)r->stage + off[i], srcs[i], sizes[i], hipMemcpyHostToDevice, st));
    AppendArgs a;
    memset(&a, 0, sizeof a);
    char *b = (char *)r->stage;
    a.game_len = (const int *)(b + off[0]);
    a.game_ret0 = (const float *)(b + off[1]);
    a.states = (const uint64_t *)(b + off[2]);
    a.move = (const uint16_t *)(b + off[3]);
    a.child_action = (const uint16_t *)(b + off[4]);
    a.nchild = (const uint8_t *)(b + off[5]);
    a.child_visits = (const uint32_t *)(b + off[6]);
    a.value = (const double *)(b + off[7]);
    a.n_src_games = (int)ng;
    a.max_plies = (int)mp;
    a.maxc = (int)mc;
    a.start_ply = start_ply;
    a.on_policy = 0; // host views carry their value targets (az_engine_export fills on-policy z)
    std::vector<int32_t> lens(v->game_len, v->game_len + ng);
    return append_common(r, a, lens, st);
}

extern "C" int az_replay_append_device(az_replay *r, const void *dev_buf, int64_t n_games, int32_t start_ply, void *stream) {
    if (!r || !dev_buf || n_games < 1 || start_ply < 0) return AZ_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    RCHK(r, hipSetDevice(r->cfg.device));
    size_t ng = (size_t)n_games, mp = (size_t)r->max_plies, mc = (size_t)r->maxc;
    const size_t sizes[8] = {ng * 4, ng * 4, ng * mp * 16, ng * mp * 2, ng * mp, ng * mp * mc * 2, ng * mp * mc * 4, ng * mp * 8};
    size_t off[9] = {0};
    for (int i = 0; i < 8; i++) off[i + 1] = off[i] + ((sizes[i] + 15) & ~(size_t)15);
    const char *b = (const char *)dev_buf;
    AppendArgs a;
    memset(&a, 0, sizeof a);
    a.game_len = (const int *)(b + off[0]);
    a.game_ret0 = (const float *)(b + off[1]);
    a.states = (const uint64_t *)(b + off[2]);
    a.move = (const uint16_t *)(b + off[3]);
    a.nchild = (const uint8_t *)(b + off[4]);
    a.child_action = (const uint16_t *)(b + off[5]);
    a.child_visits = (const uint32_t *)(b + off[6]);
    a.value = (const double *)(b + off[7]);
    a.n_src_games = (int)ng;
    a.max_plies = (int)mp;
    a.maxc = (int)mc;
    a.start_ply = start_ply;
    a.on_policy = 0; // az_engine_export_device has filled the on-policy targets in
    std::vector<int32_t> lens(ng);
    RCHK(r, hipMemcpyAsync(lens.data(), a.game_len, ng * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    RCHK(r, hipStreamSynchronize(st));
    return append_common(r, a, lens, st);
}

// ------------------------------------------------------------------------------------------------ dedupe
__global__ void gather_keys_kernel(const uint64_t *key, long long head, long long cap, long long n, uint64_t *out, long long *idx) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = key[(head + i) % cap];
    idx[i] = i;
}
__global__ void seg_flags_kernel(const uint64_t *skey, long long n, unsigned char *flag) {
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flag[j] = (j == 0 || skey[j] != skey[j - 1]) ? 1 : 0;
}
// one wave per segment: members sidx[seg_start[s] .. seg_start[s+1]) are in buffer order (stable sort)
__global__ void seg_average_kernel(const long long *seg_start, long long n_seg, long long n, const long long *sidx, long long head,
                                   long long cap, int A, double *pi, double *z, unsigned char *first_flag, const uint64_t *key2,
                                   const uint64_t *bb0, const uint64_t *bb1, const int32_t *ply, unsigned int *faults) {
    long long s = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (s >= n_seg) return;
    long long j0 = seg_start[s], j1 = s + 1 < n_seg ? seg_start[s + 1] : n;
    long long first = sidx[j0];
    if (lane == 0) first_flag[first] = 1;
    long long cnt = j1 - j0;
    if (cnt == 1) return; // x / 1 == x: nothing to write
    long long pf = (head + first) % cap;
    { // exact-key guard: every member must be the same history as the first (same fingerprint, ply and position)
        bool bad = false;
        for (long long j = j0 + 1 + lane; j < j1; j += 64) {
            long long pm = (head + sidx[j]) % cap;
            bad |= key2[pm] != key2[pf] || ply[pm] != ply[pf] || bb0[pm] != bb0[pf] || bb1[pm] != bb1[pf];
        }
        if (__ballot(bad) != 0) { // (wave-uniform) the segment is left untouched: nothing is averaged across different histories
            if (lane == 0) atomicOr(faults, AZ_REPLAY_FAULT_KEY_COLLISION);
            return;
        }
    }
    for (int a = lane; a < A; a += 64) { // flattened_buffer_dict[key][2] = [sum(x) for x in zip(acc, item[2])]
        double acc = pi[(size_t)pf * A + a];
        for (long long j = j0 + 1; j < j1; j++) acc = acc + pi[(size_t)((head + sidx[j]) % cap) * A + a];
        pi[(size_t)pf * A + a] = acc / (double)cnt;
    }
    if (lane == 0) { // flattened_buffer_dict[key][3] += item[3]
        double acc = z[pf];
        for (long long j = j0 + 1; j < j1; j++) acc += z[(head + sidx[j]) % cap];
        z[pf] = acc / (double)cnt;
    }
}
__global__ void clear_fault_kernel(unsigned int *faults, unsigned int mask) { atomicAnd(faults, ~mask); }
__global__ void iota_kernel(long long *p, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

extern "C" int az_replay_dedupe(az_replay *r, void *stream) {
    if (!r) return AZ_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    RCHK(r, hipSetDevice(r->cfg.device));
    long long n = r->n;
    r->n_unique = 0;
    // the collision flag describes THIS pass (an earlier one may have tripped on examples that have since been evicted)
    hipLaunchKernelGGL(clear_fault_kernel, dim3(1), dim3(1), 0, st, r->faults, AZ_REPLAY_FAULT_KEY_COLLISION);
    if (n == 0) return AZ_OK;
    uint64_t *k_in = nullptr, *k_out = nullptr;
    long long *i_in = nullptr, *i_out = nullptr, *seg_start = nullptr, *iota = nullptr, *d_count = nullptr;
    unsigned char *flag = nullptr, *first_flag = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(i_in); (void)hipFree(i_out); (void)hipFree(seg_start);
        (void)hipFree(iota); (void)hipFree(d_count); (void)hipFree(flag); (void)hipFree(first_flag); (void)hipFree(tmp);
    };
#define DCHK(call)                                                         \
    do {                                                                   \
        hipError_t _s = (call);                                            \
        if (_s != hipSuccess) {                                            \
            r->err = std::string(#call) + ": " + hipGetErrorString(_s);    \
            cleanup();                                                     \
            return AZ_E_HIP;                                               \
        }                                                                  \
    } while (0)
    DCHK(hipMalloc((void **)&k_in, n * 8)); DCHK(hipMalloc((void **)&k_out, n * 8));
    DCHK(hipMalloc((void **)&i_in, n * 8)); DCHK(hipMalloc((void **)&i_out, n * 8));
    DCHK(hipMalloc((void **)&seg_start, n * 8)); DCHK(hipMalloc((void **)&iota, n * 8));
    DCHK(hipMalloc((void **)&d_count, 8)); DCHK(hipMalloc((void **)&flag, n)); DCHK(hipMalloc((void **)&first_flag, n));
    unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(gather_keys_kernel, dim3(nb), dim3(256), 0, st, r->key, (long long)r->head, (long long)r->cap, n, k_in, i_in);
    size_t tb = 0;
    DCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, i_in, i_out, (int)n, 0, 64, st));
    DCHK(hipMalloc(&tmp, tb));
    DCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k_in, k_out, i_in, i_out, (int)n, 0, 64, st)); // LSD radix sort: stable
    hipLaunchKernelGGL(seg_flags_kernel, dim3(nb), dim3(256), 0, st, k_out, n, flag);
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, st, iota, n);
    size_t tb2 = 0;
    DCHK(hipcub::DeviceSelect::Flagged(nullptr, tb2, iota, flag, seg_start, d_count, (int)n, st));
    if (tb2 > tb) {
        (void)hipFree(tmp);
        tmp = nullptr;
        DCHK(hipMalloc(&tmp, tb2));
        tb = tb2;
    }
    DCHK(hipcub::DeviceSelect::Flagged(tmp, tb2, iota, flag, seg_start, d_count, (int)n, st)); // segment start positions
    long long n_seg = 0;
    DCHK(hipMemcpyAsync(&n_seg, d_count, 8, hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    DCHK(hipMemsetAsync(first_flag, 0, n, st));
    hipLaunchKernelGGL(seg_average_kernel, dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0, st, seg_start, n_seg, n, i_out,
                       (long long)r->head, (long long)r->cap, r->A, r->pi, r->z, first_flag, r->key2, r->bb0, r->bb1, r->ply, r->faults);
    DCHK(hipcub::DeviceSelect::Flagged(nullptr, tb2, iota, first_flag, (long long *)r->unique, d_count, (int)n, st));
    if (tb2 > tb) {
        (void)hipFree(tmp);
        tmp = nullptr;
        DCHK(hipMalloc(&tmp, tb2));
    }
    DCHK(hipcub::DeviceSelect::Flagged(tmp, tb2, iota, first_flag, (long long *)r->unique, d_count, (int)n, st)); // dict order
    long long n_unique = 0;
    unsigned int faults = 0;
    DCHK(hipMemcpyAsync(&n_unique, d_count, 8, hipMemcpyDeviceToHost, st));
    DCHK(hipMemcpyAsync(&faults, r->faults, 4, hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    DCHK(hipGetLastError());
    cleanup();
#undef DCHK
    if (n_unique != n_seg) {
        r->err = "internal: unique count mismatch";
        return AZ_E_DEVICE;
    }
    if (faults & AZ_REPLAY_FAULT_KEY_COLLISION) {
        r->err = "remove_duplicates: two different histories share a 64-bit key (they would have been averaged); not deduplicated";
        return AZ_E_DEVICE;
    }
    r->n_unique = n_unique;
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------ sampling
struct SampleArgs {
    AzGeom geom;
    int game, A, batch, planes_elems;
    long long head, cap, n_unique;
    const int64_t *unique, *indices;
    uint64_t seed, call;
    const uint64_t *bb0, *bb1;
    const int32_t *ply;
    const double *z, *pi;
    float *x, *pio, *zo;
    unsigned int *faults;
};
__global__ void replay_sample_kernel(SampleArgs a) {
    int b = blockIdx.x;
    long long u;
    if (a.indices) u = a.indices[b];
    else { // np.random.randint(len(flattened_buffer)) stand-in: one splitmix draw per row
        uint64_t h = mix64(mix64(a.seed, a.call), (uint64_t)b);
        u = (long long)__umul64hi(h, (uint64_t)a.n_unique);
    }
    if (u < 0 || u >= a.n_unique) { // an index outside the de-duplicated list: poison the row and raise a fault (reported
                                     // by the next az_replay_stats_get) instead of silently substituting example 0
        if (threadIdx.x == 0) {
            atomicOr(a.faults, AZ_REPLAY_FAULT_BAD_INDEX);
            a.zo[b] = __builtin_nanf("");
        }
        for (int i = threadIdx.x; i < a.planes_elems; i += blockDim.x) a.x[(size_t)b * a.planes_elems + i] = __builtin_nanf("");
        for (int i = threadIdx.x; i < a.A; i += blockDim.x) a.pio[(size_t)b * a.A + i] = __builtin_nanf("");
        return;
    }
    long long phys = (a.head + a.unique[u]) % a.cap;
    AzState s;
    s.bb0 = a.bb0[phys];
    s.bb1 = a.bb1[phys];
    s.ply = a.ply[phys];
    for (int i = threadIdx.x; i < a.planes_elems; i += blockDim.x)
        a.x[(size_t)b * a.planes_elems + i] = a.game == AZG_CONNECT_FOUR ? az_obs_elem<AZG_CONNECT_FOUR>(s, a.geom, i)
                                                                         : az_obs_elem<AZG_BREAKTHROUGH>(s, a.geom, i);
    for (int i = threadIdx.x; i < a.A; i += blockDim.x) a.pio[(size_t)b * a.A + i] = (float)a.pi[(size_t)phys * a.A + i];
    if (threadIdx.x == 0) a.zo[b] = (float)a.z[phys];
}

extern "C" int az_replay_sample(az_replay *r, const int64_t *indices, int32_t batch, uint64_t seed, float *x, float *pi,
                                float *z, void *stream) {
    if (!r || !x || !pi || !z || batch < 1) return AZ_E_INVALID;
    if (r->n_unique < 1) {
        r->err = "az_replay_sample before az_replay_dedupe (or the buffer is empty)";
        return AZ_E_STATE;
    }
    SampleArgs a;
    a.geom = r->geom;
    a.game = r->cfg.game;
    a.A = r->A;
    a.batch = batch;
    a.planes_elems = 4 * r->cfg.rows * r->cfg.cols;
    a.head = r->head;
    a.cap = r->cap;
    a.n_unique = r->n_unique;
    a.unique = r->unique;
    a.indices = indices;
    a.seed = seed;
    a.call = r->sample_calls++;
    a.bb0 = r->bb0;
    a.bb1 = r->bb1;
    a.ply = r->ply;
    a.z = r->z;
    a.pi = r->pi;
    a.x = x;
    a.pio = pi;
    a.zo = z;
    a.faults = r->faults;
    RCHK(r, hipSetDevice(r->cfg.device));
    hipLaunchKernelGGL(replay_sample_kernel, dim3(batch), dim3(128), 0, (hipStream_t)stream, a);
    RCHK(r, hipGetLastError());
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------ read-back
extern "C" int64_t az_replay_read_unique(az_replay *r, int64_t max_n, uint64_t *key, double *pi, double *z, int64_t *buffer_index,
                                         uint64_t *bitboards, int32_t *ply) {
    if (!r || max_n < 0) return AZ_E_INVALID;
    RCHK(r, hipSetDevice(r->cfg.device));
    RCHK(r, hipDeviceSynchronize());
    int64_t n = r->n_unique < max_n ? r->n_unique : max_n;
    std::vector<int64_t> u((size_t)n);
    if (n) RCHK(r, hipMemcpy(u.data(), r->unique, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) {
        int64_t phys = (r->head + u[(size_t)i]) % r->cap;
        if (buffer_index) buffer_index[i] = u[(size_t)i];
        if (key) RCHK(r, hipMemcpy(key + i, r->key + phys, 8, hipMemcpyDeviceToHost));
        if (z) RCHK(r, hipMemcpy(z + i, r->z + phys, 8, hipMemcpyDeviceToHost));
        if (pi) RCHK(r, hipMemcpy(pi + i * r->A, r->pi + (size_t)phys * r->A, 8 * (size_t)r->A, hipMemcpyDeviceToHost));
        if (bitboards) {
            RCHK(r, hipMemcpy(bitboards + 2 * i, r->bb0 + phys, 8, hipMemcpyDeviceToHost));
            RCHK(r, hipMemcpy(bitboards + 2 * i + 1, r->bb1 + phys, 8, hipMemcpyDeviceToHost));
        }
        if (ply) RCHK(r, hipMemcpy(ply + i, r->ply + phys, 4, hipMemcpyDeviceToHost));
    }
    return r->n_unique;
}

extern "C" int az_replay_debug_set_key(az_replay *r, int64_t index, uint64_t key) {
    if (!r || index < 0 || index >= r->n) return AZ_E_INVALID;
    RCHK(r, hipSetDevice(r->cfg.device));
    RCHK(r, hipDeviceSynchronize());
    RCHK(r, hipMemcpy(r->key + (r->head + index) % r->cap, &key, 8, hipMemcpyHostToDevice));
    return AZ_OK;
}

extern "C" int az_replay_read_example(az_replay *r, int64_t index, double *pi, double *z) {
    if (!r || index < 0 || index >= r->n) return AZ_E_INVALID;
    RCHK(r, hipSetDevice(r->cfg.device));
    RCHK(r, hipDeviceSynchronize());
    int64_t phys = (r->head + index) % r->cap;
    if (pi) RCHK(r, hipMemcpy(pi, r->pi + (size_t)phys * r->A, 8 * (size_t)r->A, hipMemcpyDeviceToHost));
    if (z) RCHK(r, hipMemcpy(z, r->z + phys, 8, hipMemcpyDeviceToHost));
    return AZ_OK;
}
