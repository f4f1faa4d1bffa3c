// az_engine.hip — MI355X (gfx950) self-play engine: batched PUCT tree search + rollout loop.
//
// One 64-lane wavefront owns one concurrent game ("slot").  Per tick (az_engine_advance) a slot
//   consumes the network result of its outstanding request  -> expand + backup   (mcts.py:54-66,82-89,145-152)
//   then runs playouts: select by wave argmax of PUCT          (mcts.py:38-52,68-80,139-142)
//   applying bitboard moves until it needs the network again    (az_games.h)
//   and, every S playouts, the agent's move step               (mcts.py:155-162,192-203; alphazerobot.py:71-93;
//                                                               game_utils.py:156-197)
// Tree arithmetic is IEEE double with one rounding per operation (-ffp-contract=off), i.e. the
// reference's Python-float arithmetic: given the same (priors, value) inputs and the same random
// draws, visit counts and Q values are bit-identical to mcts.py.
//
// HBM layout: node pools of 32-byte records (AzNode, az_engine_internal.h), index = pool*cap + node:
//   N   u32   visit count                       Q  f64  mean value (viewpoint of the player who moved in)
//   C0  u32   index of first child              P  f64  prior
//   META u32  action (low 16) | n_children<<16  (children are contiguous, ascending action = dict insertion order of
//                                                mcts.py:62-64)
// so one select level is ONE contiguous read of nodes [c0 .. c0+n) by lanes 0..n-1 (two 16-byte loads per lane).
// A slot owns one pool; re-rooting compacts the kept subtree into a spare pool taken from a shared set and gives the
// old pool back (Cheney copy, breadth-first) when the free tail could not hold another search.
// One launch per tick (az_advance_kernel): the agent's move step is the cold prologue of the slot's next tick.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/az_engine.h"
#include "az_games.h"

#include "az_engine_internal.h"

// ------------------------------------------------------------------------------------------------
// wave helpers
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rflu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    uint32_t lo = rflu((uint32_t)v), hi = rflu((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        double o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}
__device__ __forceinline__ unsigned int wave_or(unsigned int v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v |= (unsigned int)__shfl_xor((int)v, off);
    return v;
}
__device__ __forceinline__ uint64_t lanes_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG: stream = (seed, game id), counter = (ply, purpose, index, draw)
struct Philox {
    uint32_t k0, k1, c0, c1, c2, c3;
    uint32_t out[4];
    int have;
};
__device__ __forceinline__ void philox_init(Philox &r, uint64_t seed, uint32_t gid, uint32_t ply, uint32_t purpose, uint32_t idx) {
    r.k0 = (uint32_t)seed;
    r.k1 = (uint32_t)(seed >> 32);
    r.c0 = 0;
    r.c1 = idx;
    r.c2 = (ply << 8) | purpose;
    r.c3 = gid;
    r.have = 0;
}
__device__ __forceinline__ void philox_block(Philox &r) {
    uint32_t c0 = r.c0, c1 = r.c1, c2 = r.c2, c3 = r.c3, k0 = r.k0, k1 = r.k1;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r.out[0] = c0; r.out[1] = c1; r.out[2] = c2; r.out[3] = c3;
    r.c0++;
    r.have = 2;
}
__device__ __forceinline__ double philox_u01(Philox &r) { // [0,1), 53 bits
    if (!r.have) philox_block(r);
    r.have--;
    uint64_t x = ((uint64_t)r.out[2 * r.have] << 32) | r.out[2 * r.have + 1];
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
// Gamma(alpha) by Marsaglia-Tsang (alpha < 1 boosted by U^(1/alpha)).  Only the distribution matters here (production
// noise, checked statistically), so the transcendental work runs in fp32 fast math: the fp64 log/pow/cos of the first
// version made the cold move kernel average 15 us per tick.
__device__ double philox_gamma(Philox &r, double alpha) {
    float a = (float)alpha, boost = 1.f;
    if (a < 1.f) {
        boost = __powf(1.f - (float)philox_u01(r), 1.f / a);
        a += 1.f;
    }
    float d = a - 1.f / 3.f, c = rsqrtf(9.f * d);
    for (int it = 0; it < 64; it++) {
        float u1 = 1.f - (float)philox_u01(r), u2 = (float)philox_u01(r);
        float x = sqrtf(-2.f * __logf(fmaxf(u1, 1e-30f))) * __cosf(6.2831853f * u2);
        float v = 1.f + c * x;
        if (v <= 0.f) continue;
        v = v * v * v;
        float u = 1.f - (float)philox_u01(r);
        if (__logf(fmaxf(u, 1e-30f)) < 0.5f * x * x + d - d * v + d * __logf(v)) return (double)fmaxf(d * v * boost, 1e-30f);
    }
    return (double)fmaxf(d * boost, 1e-30f);
}

// ------------------------------------------------------------------------------------------------
// numpy arithmetic on the agent's move step (alphazerobot.py:13-14,78,84), lanes = root children.
__device__ __forceinline__ double np_pow(double x, double e) { // ndarray ** python-float fast paths
    if (e == 1.0) return x;
    if (e == 2.0) return x * x;
    if (e == 0.5) return sqrt(x);
    if (e == -1.0) return 1.0 / x;
    return pow(x, e);
}

// np.sum (pairwise, 8 accumulators per <=128 block) of the dense length-A vector whose only non-zeros
// are v[k] at index act[k], k < nc, ascending.  Executed redundantly by all lanes (uniform control).
__device__ double np_sum_sparse(const PwPlan &pw, double v, int act, int nc) {
    double stack[6];
    int sp = 0;
    int k = 0; // children are consumed in ascending index order across ascending blocks
    for (int o = 0; o < pw.n_ops; o++) {
        int op = pw.ops[o];
        if (op < 0) {
            double b = stack[--sp], a = stack[--sp];
            stack[sp++] = a + b;
            continue;
        }
        int lo = pw.lo[op], n = pw.len[op];
        double res;
        if (n < 8) {
            res = 0.0;
            while (k < nc) {
                int a = __shfl(act, k);
                if (a >= lo + n) break;
                res += __shfl(v, k);
                k++;
            }
        } else {
            double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, tail = 0;
            int body = n - (n % 8);
            int k0 = k;
            while (k < nc) {
                int a = __shfl(act, k);
                if (a >= lo + n) break;
                int j = a - lo;
                double x = __shfl(v, k);
                if (j < body) {
                    switch (j & 7) {
                    case 0: r0 += x; break;
                    case 1: r1 += x; break;
                    case 2: r2 += x; break;
                    case 3: r3 += x; break;
                    case 4: r4 += x; break;
                    case 5: r5 += x; break;
                    case 6: r6 += x; break;
                    default: r7 += x; break;
                    }
                }
                k++;
            }
            res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
            for (int kk = k0; kk < k; kk++) { // remainder elements are added after the tree, in order
                int j = __shfl(act, kk) - lo;
                if (j >= body) res += __shfl(v, kk);
            }
            (void)tail;
        }
        stack[sp++] = res;
    }
    return stack[0];
}

// ------------------------------------------------------------------------------------------------
struct Pool {
    AzNode *nd;
};
#define POOL_MASK 0x3FFFFFFF // which[g] = pool id | select rule << 30
__device__ __forceinline__ Pool pool_at(const Params &p, int pool) {
    Pool q = {p.nodes + (size_t)pool * p.cap};
    return q;
}
__device__ __forceinline__ void pool_init_root(const Pool &t, int lane) {
    if (lane == 0) {
        AzNode r = {0u, NONE32, 0u, 0u, 0.0, 0.0};
        t.nd[0] = r;
    }
}

template <int NP> struct Path { // depth d lives in lane d&63, register d>>6
    uint32_t r[NP];  // node index
    uint32_t pn[NP]; // the node's N and Q as last read or written by this wave: the backup needs no second read
    double pq[NP];
    __device__ __forceinline__ void set(int lane, int d, uint32_t node, uint32_t n, double q) {
#pragma unroll
        for (int i = 0; i < NP; i++)
            if ((d >> 6) == i && lane == (d & 63)) {
                r[i] = node;
                pn[i] = n;
                pq[i] = q;
            }
    }
};
__device__ __forceinline__ double readlane_d(double v, int src) { // src wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Enumerate the legal actions of `s` in ascending order, one wave.  For each of this lane's (<=3)
// moves: slot index k (position in the ascending list) and action id.  Returns total count.
template <int GAME>
__device__ __forceinline__ int enum_moves(const AzState &s, const AzGeom &g, int lane, int k[3], int act[3], int &mine) {
    if (GAME == AZG_CONNECT_FOUR) {
        uint32_t m = az_c4_legal_mask(s);
        bool ok = lane < 7 && ((m >> lane) & 1u);
        mine = ok ? 1 : 0;
        k[0] = __popc(m & ((1u << lane) - 1u));
        act[0] = lane;
        k[1] = k[2] = act[1] = act[2] = 0;
        return __popc(m);
    } else {
        uint32_t mv = lane < g.cells ? az_bt_cell_moves(s, g, lane) : 0u;
        uint64_t b0 = __ballot(mv & 1u), b1 = __ballot(mv & 2u), b2 = __ballot(mv & 4u);
        uint64_t below = lanes_below(lane);
        int off = __popcll(b0 & below) + __popcll(b1 & below) + __popcll(b2 & below);
        int me = s.ply & 1;
        // the lane's j-th move = the j-th set direction bit.  Every index below is a compile-time constant: an array indexed
        // by a run-time value (k[mine++]) lives in scratch memory, and its round trips sat on the expansion's critical path
        // (measured with clock64 probes: 22.6 k cycles per tick for breakthrough against 3.4 k for connect_four)
        uint32_t m3 = mv & 7u;
        mine = __popc(m3);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int d = m3 ? __ffs(m3) - 1 : 0;
            k[j] = off + j;
            act[j] = az_bt_encode(lane, me, d, (mv >> (4 + d)) & 1u);
            m3 &= m3 - 1u;
        }
        return __popcll(b0) + __popcll(b1) + __popcll(b2);
    }
}

template <int GAME> __device__ __forceinline__ void write_obs(const Params &p, const AzState &s, float *obs, int lane) {
    for (int i = lane; i < p.obs_elems; i += 64) obs[i] = az_obs_elem<GAME>(s, p.geom, i);
}

// Cheney copy of the subtree under `root` from pool `a` into pool `b` (node 0 = new root).
// Returns the number of live nodes.  Children stay contiguous and in ascending-action order.
__device__ uint32_t compact_subtree(const Pool &a, const Pool &b, uint32_t root, int lane) {
    if (lane == 0) {
        b.nd[0] = a.nd[root]; // C0 is still an OLD index until scanned
    }
    __threadfence_block();
    uint32_t s = 0, f = 1;
    while (s < f) {
        uint32_t cnt = f - s < 64 ? f - s : 64;
        uint32_t oc0 = NONE32;
        int nch = 0;
        if ((uint32_t)lane < cnt) {
            oc0 = b.nd[s + lane].C0;
            nch = oc0 == NONE32 ? 0 : (int)(b.nd[s + lane].META >> 16);
        }
        int incl = wave_incl_scan(nch, lane);
        int total = __shfl(incl, 63);
        uint32_t dst = f + (uint32_t)(incl - nch);
        if (nch > 0) b.nd[s + lane].C0 = dst;
        int mx = nch;
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            int o = __shfl_xor(mx, off);
            mx = o > mx ? o : mx;
        }
        // (the two pool halves never overlap: batch the loads of four children before their stores, so that four HBM round
        //  trips overlap instead of one load -> store -> load chain per child)
        for (int k0 = 0; k0 < mx; k0 += 4)
            if (k0 < nch) { // a node = two 16-byte words; clamped, unconditional loads keep the four nodes in registers
                const int last = nch - 1;
                const uint4 *s0 = (const uint4 *)(a.nd + oc0 + k0), *s1 = (const uint4 *)(a.nd + oc0 + (k0 + 1 < last ? k0 + 1 : last)),
                            *s2 = (const uint4 *)(a.nd + oc0 + (k0 + 2 < last ? k0 + 2 : last)),
                            *s3 = (const uint4 *)(a.nd + oc0 + (k0 + 3 < last ? k0 + 3 : last));
                const uint4 a0 = s0[0], b0 = s0[1], a1 = s1[0], b1 = s1[1], a2 = s2[0], b2 = s2[1], a3 = s3[0], b3 = s3[1];
                uint4 *d = (uint4 *)(b.nd + dst + k0);
                d[0] = a0;
                d[1] = b0;
                if (k0 + 1 < nch) {
                    d[2] = a1;
                    d[3] = b1;
                }
                if (k0 + 2 < nch) {
                    d[4] = a2;
                    d[5] = b2;
                }
                if (k0 + 3 < nch) {
                    d[6] = a3;
                    d[7] = b3;
                }
            }
        __threadfence_block(); // the next chunk reads what this one wrote (same wave, global memory)
        f += (uint32_t)total;
        s += cnt;
    }
    return f;
}

// ------------------------------------------------------------------------------------------------
// Slot registers shared by the kernels
struct SlotRegs {
    int gid, sims, pool;
    int rule; // AZ_SELECT_* of the slot's current tree (the reference's Node.use_puct, uniform within a tree)
    uint32_t root, alloc;
    AzState rs;
};
__device__ __forceinline__ void slot_load(const Params &p, int g, SlotRegs &r) {
    r.gid = rfl(p.gid[g]);
    r.rs.bb0 = rfl64(p.bb0[g]);
    r.rs.bb1 = rfl64(p.bb1[g]);
    r.rs.ply = rfl(p.ply[g]);
    r.sims = rfl(p.sims[g]);
    r.pool = rfl(p.which[g]);
    r.rule = (r.pool >> 30) & 1;
    r.pool &= POOL_MASK;
    r.root = rflu(p.root[g]);
    r.alloc = rflu(p.alloc[g]);
}
__device__ __forceinline__ void slot_store(const Params &p, int g, const SlotRegs &r, int phase) {
    p.phase[g] = phase;
    p.gid[g] = r.gid;
    p.bb0[g] = r.rs.bb0;
    p.bb1[g] = r.rs.bb1;
    p.ply[g] = r.rs.ply;
    p.sims[g] = r.sims;
    p.which[g] = r.pool | (r.rule << 30);
    p.root[g] = r.root;
    p.alloc[g] = r.alloc;
}

// mcts.py:82-89 update_recursive along the recorded path: node at depth d gets x * (-1)^(depth-d).  N and Q of every
// path node are already in the lane's registers (read on the way down, or loaded with the slot), so this only stores.
template <int NP>
__device__ __forceinline__ void backup_path(const Pool &t, Path<NP> &path, int depth, double x, int lane) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
        int d = i * 64 + lane;
        if (d <= depth) {
            uint32_t nd = path.r[i];
            double xv = ((depth - d) & 1) ? -x : x;
            uint32_t nn = path.pn[i];
            double q = ((double)nn * path.pq[i] + xv) / (double)(nn + 1); // mcts.py:83
            t.nd[nd].Q = q;
            t.nd[nd].N = nn + 1;
            path.pq[i] = q;
            path.pn[i] = nn + 1;
        }
    }
}

// mcts.update_root(action) (mcts.py:192-203) + the pool bookkeeping it implies here.
// `sel` = index of the chosen child among the root's children, or -1 when the action is not among them (leaf root).
// `drop`: the tree is dropped whatever it holds (a new game, or keep_search_tree=False: alphazerobot.py:66-67).
// `fresh_rule`: the select rule of a root created here.  The reference keeps it per Node (use_puct, inherited by children:
// mcts.py:64): MCTS.__init__'s root is always PUCT (mcts.py:122), a leaf root replaced by update_root takes MCTS.use_puct
// (mcts.py:199-200) - so the caller says which of the two the reference would have built at this point.
__device__ void reroot(const Params &p, int g, SlotRegs &sr, Pool &t, int sel, bool drop, int fresh_rule, int lane,
                       unsigned int &fault, unsigned long long &st_compact, int rsv_k = -1, int rsv_np = -1) {
    uint32_t c0 = rflu(t.nd[sr.root].C0);
    if (drop || sel < 0 || c0 == NONE32) {
        sr.rule = fresh_rule;
        sr.root = 0;
        sr.alloc = 1;
        pool_init_root(t, lane);
        __threadfence_block();
        return;
    }
    sr.root = c0 + (uint32_t)sel;
    if (sr.alloc + p.need_per_move > p.cap) {
        // the target is a spare pool: entry k of spare[] holds a free pool id or -1.  In a launch with deferred compaction the
        // caller has RESERVED one before it changed anything (move_step); otherwise take one here - every taker hands a pool
        // back (its old one, into the entry it emptied) as soon as its copy is done, so a waiting wave waits for copies in progress.
        int k = rsv_k, np = rsv_np;
        if (np < 0 && lane == 0) {
            int kk = g % p.n_spare;
            for (int spin = 0; spin < (1 << 22); spin++) {
                int v = atomicExch(&p.spare[kk], -1);
                if (v >= 0) {
                    k = kk;
                    np = v;
                    break;
                }
                kk = kk + 1 == p.n_spare ? 0 : kk + 1;
                __builtin_amdgcn_s_sleep(4);
            }
        }
        k = __shfl(k, 0);
        np = __shfl(np, 0);
        if (np < 0) { // (no copy finished within ~a second: cannot happen unless the device is wedged)
            fault |= AZ_FAULT_POOL_EXHAUSTED;
            return;
        }
        st_compact++;
        if (rsv_np >= 0) {
            // the copy goes to the extra workgroups of THIS launch (compact_jobs: a workgroup per job; it writes alloc[g]).  The
            // caller publishes the job once the slot's state is stored (move_step); nothing in the rest of this tick touches the
            // tree, and the next tick finds it at node 0 of the new pool
            if (lane == 0) {
                p.cj_from[g] = sr.pool;
                p.cj_entry[g] = k;
                p.cj_root[g] = sr.root;
            }
            sr.root = 0;
            sr.pool = np;
            sr.alloc = p.cap; // (placeholder: compact_jobs stores the live count)
            t = pool_at(p, np);
            return;
        }
        Pool o = pool_at(p, np);
        sr.alloc = compact_subtree(t, o, sr.root, lane);
        // (every load from the old pool has returned - its data went into the stores above - so the pool can change hands)
        if (lane == 0) atomicExch(&p.spare[k], sr.pool);
        sr.root = 0;
        sr.pool = np;
        t = o;
        if (sr.alloc + p.need_per_move > p.cap) fault |= AZ_FAULT_POOL_EXHAUSTED;
    }
}

// Handed-over compaction: the last AZ_COMPACT_WGS workgroups of a tick kernel launch take the copies the slot waves of the SAME launch
// hand over (move_step: the slot's state is stored, then cj_job[row] = slot + 1 with release semantics), beside the other slots'
// playouts.  One 256-thread workgroup per job copies the subtree under cj_root[g] of pool cj_from[g] into the slot's new pool,
// breadth first, 256 parents per round: the children of a round are numbered by a block-wide prefix sum (the SAME order - hence
// the same node indices - as compact_subtree's wave-wide one) and copied one node per thread and step, so a round is two or three
// memory round trips whatever the fan-out.
// Who serves whom, and when a workgroup may leave: extra workgroup k owns rows [k R, (k + 1) R) of the launch (R = rows / workgroups,
// rounded up).  Every slot wave ends by storing the launch's epoch into cj_seen[row] (a plain agent-scope store to an address of
// its own - a shared counter of finished waves, 2048 atomic adds on one address per launch, added 20 us to every launch); the
// workgroup polls its rows' cj_job and cj_seen (one or two coalesced loads a microsecond) and leaves when all of them have been
// seen and none has a job - so the launch ends with every job done, and a slot never carries a pending copy across launches: a
// captured graph may hold any number of launches, and the other entry points (advance_slots, read_tree, ...) never meet a
// half-copied pool.  The epoch is a device counter the last extra workgroup out advances: constant while a launch runs (nothing
// host-side that a graph would freeze).  (The copies ran in a kernel of their own for a day: 4.9 us per tick, empty or not; then
// in the NEXT launch, keyed by a host-side parity that a graph of an odd number of launches froze: ADVICE r3.)
#define CJ_THREADS 256
#define AZ_COMPACT_WGS 64 // extra workgroups per launch (jobs are rare; a burst is worked off 64 at a time)
__device__ void compact_jobs(const Params &p, const int wg, const int n_wg, const int n_rows) {
    __shared__ uint32_t sh_off[CJ_THREADS + 1], sh_c0[CJ_THREADS];
    __shared__ int sh_wave[CJ_THREADS / 64];
    __shared__ int sh_job, sh_unseen;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int epoch = p.cjob_count[0];
    const int per = (n_rows + n_wg - 1) / n_wg, r0 = wg * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
    bool confirmed = false;
    for (int spin = 0; r0 < r1 && spin < (1 << 20); spin++) { // (bounded: ~a second; a launch lasts tens of microseconds)
        if (tid == 0) sh_job = -1, sh_unseen = 0;
        __syncthreads();
        for (int r = r0 + tid; r < r1; r += CJ_THREADS) { // relaxed polls (an acquire per poll would invalidate this XCD's L2 under the slot waves)
            // (seen first: a wave's job store is acknowledged before its seen store is issued, so "seen and no job" is final)
            const int seen = __hip_atomic_load(p.cj_seen + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int j = __hip_atomic_load(p.cj_job + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (j > 0) sh_job = r; // (any one of them: benign race)
            if (seen != epoch) sh_unseen = 1;
        }
        __syncthreads();
        const int jr = sh_job, unseen = sh_unseen;
        __syncthreads();
        if (jr < 0) {
            if (!unseen) {
                if (confirmed) break; // (the pass before saw every row out; THIS pass's job loads were issued after those had returned)
                confirmed = true;
                continue;
            }
            __builtin_amdgcn_s_sleep(48);
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // ONE acquire per job: what the publisher stored before its release
        const int g = __hip_atomic_load(p.cj_job + jr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
        const Pool a = pool_at(p, p.cj_from[g]), b = pool_at(p, p.which[g] & POOL_MASK);
        if (tid == 0) b.nd[0] = a.nd[p.cj_root[g]]; // C0 is still an OLD index until scanned
        __threadfence_block();
        __syncthreads();
        uint32_t s = 0, f = 1;
        while (s < f) {
            const uint32_t cnt = f - s < CJ_THREADS ? f - s : CJ_THREADS;
            uint32_t oc0 = NONE32;
            int nch = 0;
            if ((uint32_t)tid < cnt) {
                oc0 = b.nd[s + tid].C0;
                nch = oc0 == NONE32 ? 0 : (int)(b.nd[s + tid].META >> 16);
            }
            int incl = wave_incl_scan(nch, lane);
            if (lane == 63) sh_wave[wv] = incl;
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < CJ_THREADS / 64; w++) {
                if (w < wv) base += sh_wave[w];
                total += sh_wave[w];
            }
            const uint32_t off = (uint32_t)(base + incl - nch); // exclusive prefix: this parent's first child among the round's
            sh_off[tid] = off;
            sh_c0[tid] = oc0;
            if (tid == CJ_THREADS - 1) sh_off[CJ_THREADS] = (uint32_t)total;
            if (nch > 0) b.nd[s + tid].C0 = f + off;
            __syncthreads();
            // child c of the round (0 <= c < total) belongs to the last parent whose prefix is <= c
            for (uint32_t c = (uint32_t)tid; c < (uint32_t)total; c += CJ_THREADS) {
                int lo = 0, hi = CJ_THREADS - 1; // largest i with sh_off[i] <= c and a non-empty range
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (sh_off[mid] <= c) lo = mid;
                    else hi = mid - 1;
                }
                const uint4 *src = (const uint4 *)(a.nd + sh_c0[lo] + (c - sh_off[lo]));
                uint4 *dst = (uint4 *)(b.nd + f + c);
                const uint4 w0 = src[0], w1 = src[1];
                dst[0] = w0;
                dst[1] = w1;
            }
            __threadfence_block(); // the next round reads what this one wrote (same workgroup, global memory)
            __syncthreads();
            f += (uint32_t)total;
            s += cnt;
        }
        if (tid == 0) {
            p.alloc[g] = f;
            __hip_atomic_store(p.cj_job + jr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (done; its own next poll must see it)
            atomicExch(&p.spare[p.cj_entry[g]], p.cj_from[g]);                               // the old pool changes hands
            if (f + p.need_per_move > p.cap) atomicOr(p.faults, AZ_FAULT_POOL_EXHAUSTED);
        }
        __syncthreads();
    }
    if (tid == 0) { // (no fence: nothing here is read before the launch ends, and an agent-scope release at this point - the XCD's L2
                    //  full of the launch's node writes - is tens of microseconds)
        if (atomicAdd(p.cjob_count + 2, 1) == n_wg - 1) { // every extra workgroup has seen its rows out and is done: the next launch's epoch
            p.cjob_count[2] = 0;
            p.cjob_count[0] = epoch + 1;
        }
    }
}

// The slot's game is over: publish its length and result, take the next game id of the generation (device counter).
// Returns false when no game is left (the slot is stored idle).
__device__ __forceinline__ bool finish_game_take_next(const Params &p, int g, int lane, SlotRegs &sr, float ret0,
                                                      unsigned long long st_moves) {
    unsigned long long nxt = 0;
    if (lane == 0) {
        p.rec_len[sr.gid] = sr.rs.ply - p.start.ply;
        p.rec_ret0[sr.gid] = ret0;
        __threadfence(); // records before the done-count
        atomicAdd(p.games_done, 1ull);
        nxt = atomicAdd(p.next_game, 1ull);
    }
    nxt = ((unsigned long long)rflu((uint32_t)(nxt >> 32)) << 32) | rflu((uint32_t)nxt);
    if ((long long)nxt >= p.n_games) {
        sr.gid = -1;
        if (lane == 0) {
            slot_store(p, g, sr, PH_IDLE);
            p.stats[(size_t)g * ST_N + ST_MOVES] += st_moves;
        }
        return false;
    }
    sr.gid = (int)nxt;
    sr.rs = p.start;
    sr.sims = 0;
    return true;
}
// Rule of the tree a game's first search runs on, when the agent is the first to move from p.start.  A bot that keeps its
// tree calls update_root on its (leaf) constructor root before its first search when the history is long enough - self-play
// bots with >= 1 move played (alphazerobot.py:57-59), the others with >= 2 (:62-64) - which installs MCTS.use_puct.
__device__ __forceinline__ int start_rule(const Params &p) {
    if (!p.keep_tree || p.manual_moves) return AZ_SELECT_PUCT;
    return p.start.ply >= (p.arena_agent == AZ_ARENA_SELF_PLAY ? 1 : 2) ? p.select_rule : AZ_SELECT_PUCT;
}
// arena: the agent plays side gid & 1; is it the OPPONENT's turn in state s of game gid?
__device__ __forceinline__ bool opponent_to_move(const Params &p, int gid, const AzState &s) {
    return p.arena_agent != AZ_ARENA_SELF_PLAY && ((s.ply ^ gid ^ p.arena_flip) & 1);
}

// ------------------------------------------------------------------------------------------------
// The cold path of a tick: AlphaZeroBot.step's tail + play_game_self's loop body for a slot whose S playouts are
// done (phase PH_MOVE, set by the previous tick), game turnover, and the root-evaluation request of the next search.
// Runs at the START of the slot's wave in az_advance_kernel and ends the wave's tick, so its registers never
// overlap the playout loop's (it used to be a kernel of its own: one more launch gap and ~11 us of serial latency
// per tick; now moving slots run beside the other slots' playouts).
template <int GAME>
__device__ __forceinline__ void move_step(const Params &p, const int g, const int lane, int ph, SlotRegs sr,
                                          float *__restrict__ obs_row, const int row) { // obs_row: row `row` of the request buffer, this slot's
    const AzGeom &geom = p.geom;
    Pool t = pool_at(p, sr.pool);
    unsigned long long st_moves = 0, st_evals = 0, st_compact = 0;
    unsigned int fault = 0;
    // Deferred compaction (compact_jobs): if this move may have to compact, RESERVE the spare pool before anything is
    // changed; a slot that finds none free (they are all with jobs queued in this launch) simply returns - its phase and state
    // are untouched, it repeats the move step next tick, after the queued copies have handed their pools back.
    int rsv_k = -1, rsv_np = -1;
    if (p.defer_compact && p.keep_tree && !p.manual_moves && (ph == PH_MOVE || ph == PH_OPP_DONE) && sr.alloc + p.need_per_move > p.cap) {
        if (lane == 0) {
            int kk = g % p.n_spare;
            for (int i = 0; i < (p.n_spare < 64 ? p.n_spare : 64); i++) {
                int v = atomicExch(&p.spare[kk], -1);
                if (v >= 0) {
                    rsv_k = kk;
                    rsv_np = v;
                    break;
                }
                kk = kk + 1 == p.n_spare ? 0 : kk + 1;
            }
        }
        rsv_k = __shfl(rsv_k, 0);
        rsv_np = __shfl(rsv_np, 0);
        if (rsv_np < 0) return;
    }
    bool queued = false; // the re-rooting of this move needs a compaction: the job is published below, after the slot's state
    auto release_rsv = [&]() { // (paths that leave the move step early hand an unused reservation back)
        if (rsv_np >= 0 && lane == 0) atomicExch(&p.spare[rsv_k], rsv_np);
        rsv_np = -1;
    };

    if (ph == PH_OPP_DONE) { // arena: apply the move the opponent bot chose; the agent's tree follows it (alphazerobot.py:60-64)
        const int action = rfl(p.opp_action[g]);
        uint32_t c0 = rflu(t.nd[sr.root].C0);
        int nc = c0 == NONE32 ? 0 : (int)(rflu(t.nd[sr.root].META) >> 16);
        int cact = lane < nc ? (int)(t.nd[c0 + lane].META & 0xFFFFu) : -1;
        unsigned long long hit = __ballot(cact == action);
        const int sel = (p.keep_tree && hit) ? __ffsll(hit) - 1 : -1;
        if (sr.rs.ply >= p.max_plies || sr.gid >= p.max_games || sr.gid < 0) fault |= AZ_FAULT_PLY_OVERFLOW;
        else if (lane == 0) { // the ply is recorded (state, move; no search statistics: n_children = 0)
            size_t ri = (size_t)sr.gid * p.max_plies + sr.rs.ply;
            p.rec_states[ri * 2] = sr.rs.bb0;
            p.rec_states[ri * 2 + 1] = sr.rs.bb1;
            p.rec_move[ri] = (uint16_t)action;
            p.rec_nchild[ri] = 0;
            p.rec_value[ri] = 0.0;
        }
        float ret0 = 0.f;
        int term = fault ? 0 : az_apply<GAME>(sr.rs, geom, action, &ret0);
        sr.sims = 0;
        if (fault) {
            ph = PH_IDLE;
        } else if (term) {
            if (!finish_game_take_next(p, g, lane, sr, ret0, 0)) {
                release_rsv();
                return;
            }
            reroot(p, g, sr, t, -1, true, start_rule(p), lane, fault, st_compact);
            ph = opponent_to_move(p, sr.gid, sr.rs) ? PH_OPPONENT : (p.use_dirichlet ? PH_NEED_ROOT : PH_RUN);
        } else { // a leaf root here = the agent's first step of the game: update_root runs only with >= 2 moves played
            reroot(p, g, sr, t, sel, !p.keep_tree, (p.keep_tree && sr.rs.ply >= 2) ? p.select_rule : AZ_SELECT_PUCT, lane, fault,
                   st_compact, rsv_k, rsv_np);
            if (sr.pool == rsv_np) {
                rsv_np = -1; // (used)
                queued = true;
            }
            ph = p.use_dirichlet ? PH_NEED_ROOT : PH_RUN;
        }
    }
    if (ph == PH_MOVE) {
        if (p.manual_moves) {
            if (lane == 0) p.phase[g] = PH_SEARCH_DONE;
            return;
        }
        uint32_t c0 = rflu(t.nd[sr.root].C0);
        int nc = c0 == NONE32 ? 0 : (int)(rflu(t.nd[sr.root].META) >> 16);
        uint32_t cn = 0;
        int cact = 0;
        double cq = 0.0;
        if (lane < nc) {
            cn = t.nd[c0 + lane].N;
            cq = t.nd[c0 + lane].Q;
            cact = (int)(t.nd[c0 + lane].META & 0xFFFFu);
        }
        long long tot = wave_sum_ll((long long)cn);
        if (nc == 0 || (tot <= 0 && p.arena_agent != AZ_ARENA_NET)) fault |= AZ_FAULT_NO_VISITS;
        if (sr.rs.ply >= p.max_plies || sr.gid >= p.max_games || sr.gid < 0) fault |= AZ_FAULT_PLY_OVERFLOW;
        if (fault) {
            release_rsv();
            if (lane == 0) {
                p.phase[g] = PH_IDLE;
                atomicOr(p.faults, fault);
            }
            return;
        }
        // value target (game_utils.py:168-194)
        double target = 0.0;
        if (p.backup == AZ_BACKUP_SOFT_Z) {
            target = -t.nd[sr.root].Q;
        } else if (p.backup == AZ_BACKUP_A0C) {
            target = wave_max(lane < nc ? (cn > 0 ? cq : -99.0) : -INFINITY);
        } else if (p.backup == AZ_BACKUP_OFF_POLICY) { // A0GB: walk the most-visited line
            uint32_t node = sr.root, nn = rflu(t.nd[sr.root].N);
            double value = 0.0, mult = 1.0;
            uint32_t kc0 = c0;
            int knc = nc;
            while (knc > 0) {
                value = t.nd[node].Q;
                double sc = -INFINITY;
                uint32_t n2 = 0, c2 = NONE32, m2 = 0;
                if (lane < knc) {
                    n2 = t.nd[kc0 + lane].N;
                    c2 = t.nd[kc0 + lane].C0;
                    m2 = t.nd[kc0 + lane].META;
                    sc = n2 > 0 ? (double)n2 + t.nd[kc0 + lane].P : -99.0;
                }
                double mx = wave_max(sc);
                int best = __ffsll((unsigned long long)__ballot(sc == mx)) - 1;
                node = kc0 + (uint32_t)best;
                nn = rflu(__shfl(n2, best));
                kc0 = rflu(__shfl(c2, best));
                knc = kc0 == NONE32 ? 0 : (int)(rflu(__shfl(m2, best)) >> 16);
                mult *= -1.0;
            }
            if (nn > 0) {
                value = t.nd[node].Q;
                mult *= -1.0;
            }
            target = value * mult;
        }
        // action sampling: visit fractions -> remove_illegal_actions -> temperature -> np.random.choice
        double nv = (lane < nc && tot > 0) ? (double)cn / (double)tot : 0.0; // mcts.py:162
        double ssum = np_sum_sparse(p.pw, nv, cact, nc);        // alphazerobot.py:13
        if (ssum > 1e-6) nv = nv / ssum; else nv = lane < nc ? 1.0 / (double)nc : 0.0;
        double ap = lane < nc ? np_pow(nv, p.inv_temp) : 0.0; // alphazerobot.py:78
        double tot2 = 0.0;
        for (int kk = 0; kk < nc; kk++) tot2 += __shfl(ap, kk);
        ap = ap / tot2;
        double run = 0.0, mycdf = 0.0; // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u,'right')
        for (int kk = 0; kk < nc; kk++) {
            run += __shfl(ap, kk);
            if (lane == kk) mycdf = run;
        }
        mycdf = mycdf / run;
        double u;
        if (p.rng_mode == AZ_RNG_INJECTED) u = p.us[(size_t)sr.gid * p.max_plies + sr.rs.ply];
        else {
            Philox r;
            philox_init(r, p.seed, (uint32_t)sr.gid, (uint32_t)sr.rs.ply, 1u, 0u);
            u = philox_u01(r);
        }
        unsigned long long gt = __ballot(lane < nc && mycdf > u);
        int sel = gt ? __ffsll(gt) - 1 : nc - 1;
        while (sel > 0 && __shfl(cn, sel) == 0) sel--; // rounding corner: never pick an unvisited child
        if ((p.arena_agent == AZ_ARENA_ZERO && !p.arena_prob) || (p.arena_agent != AZ_ARENA_NET && sr.rs.ply >= p.n_prob_plies)) {
            // outside self-play the bot is greedy unless use_probabilistic_actions, and every bot is once
            // num_probabilistic_actions moves are played: np.argmax of the tempered visit fractions (alphazerobot.py:81-86)
            // = the first most-visited child
            double mx = wave_max(lane < nc ? ap : -INFINITY);
            sel = __ffsll((unsigned long long)__ballot(lane < nc && ap == mx)) - 1;
        } else if (p.arena_agent == AZ_ARENA_NET) { // NeuralNetBot.step (alphazerobot.py:105-120): argmax of the masked,
                                                    // renormalised network priors; the children hold them after the root expansion
            double pv = lane < nc ? t.nd[c0 + lane].P : 0.0;
            double ps = np_sum_sparse(p.pw, pv, cact, nc);
            if (ps > 1e-6) pv = pv / ps; else pv = lane < nc ? 1.0 / (double)nc : 0.0;
            double mx = wave_max(lane < nc ? pv : -INFINITY);
            sel = __ffsll((unsigned long long)__ballot(lane < nc && pv == mx)) - 1;
        }
        int action = rfl(__shfl(cact, sel));
        // the example record (game_utils.py:169): state + root child visits; pi = N/sum is formed on the host
        size_t ri = (size_t)sr.gid * p.max_plies + sr.rs.ply;
        if (lane == 0) {
            p.rec_states[ri * 2] = sr.rs.bb0;
            p.rec_states[ri * 2 + 1] = sr.rs.bb1;
            p.rec_move[ri] = (uint16_t)action;
            p.rec_nchild[ri] = (uint8_t)nc;
            p.rec_value[ri] = target;
        }
        if (lane < nc) {
            p.rec_child_action[ri * p.maxc + lane] = (uint16_t)cact;
            p.rec_child_visits[ri * p.maxc + lane] = cn;
        }
        st_moves++;
        float ret0 = 0.f;
        int term = az_apply<GAME>(sr.rs, geom, action, &ret0); // game_utils.py:197
        sr.sims = 0;
        if (term) {
            if (!finish_game_take_next(p, g, lane, sr, ret0, st_moves)) {
                release_rsv();
                return;
            }
            reroot(p, g, sr, t, -1, true, start_rule(p), lane, fault, st_compact);
        } else {
            reroot(p, g, sr, t, sel, !p.keep_tree, p.keep_tree ? p.select_rule : AZ_SELECT_PUCT, lane, fault, st_compact, rsv_k, rsv_np);
            if (sr.pool == rsv_np) {
                rsv_np = -1; // (used)
                queued = true;
            }
        }
        ph = opponent_to_move(p, sr.gid, sr.rs) ? PH_OPPONENT : (p.use_dirichlet ? PH_NEED_ROOT : PH_RUN);
    }
    if (ph == PH_NEED_ROOT) { // expand_root_dirichlet's policy_fn(state) (mcts.py:183)
        if (p.rng_mode == AZ_RNG_PHILOX) { // np.random.dirichlet(0.3 * ones(n_legal)) (mcts.py:187): gamma draws / their sum
            int k[3], act[3], mine;
            enum_moves<GAME>(sr.rs, geom, lane, k, act, mine);
            double e0 = 0.0, e1 = 0.0, e2 = 0.0, part = 0.0;
            for (int j = 0; j < mine; j++) { // (a run-time loop keeps ONE copy of the sampler; selects keep k / eta in registers)
                Philox r;
                philox_init(r, p.seed, (uint32_t)sr.gid, (uint32_t)sr.rs.ply, 0u, (uint32_t)(j == 0 ? k[0] : (j == 1 ? k[1] : k[2])));
                const double x = philox_gamma(r, p.alpha);
                if (j == 0) e0 = x; else if (j == 1) e1 = x; else e2 = x;
                part += x;
            }
            double inv = 1.0 / wave_sum_d(part);
            if (mine > 0) p.eta_buf[(size_t)g * p.maxc + k[0]] = e0 * inv;
            if (mine > 1) p.eta_buf[(size_t)g * p.maxc + k[1]] = e1 * inv;
            if (mine > 2) p.eta_buf[(size_t)g * p.maxc + k[2]] = e2 * inv;
        }
        write_obs<GAME>(p, sr.rs, obs_row, lane);
        st_evals++;
        ph = PH_WAIT_ROOT;
    }
    fault = wave_or(fault);
    if (lane == 0) {
        if (rsv_np >= 0) atomicExch(&p.spare[rsv_k], rsv_np); // reserved, not needed after all (game over, fresh tree, leaf root)
        slot_store(p, g, sr, fault ? PH_IDLE : ph);
        unsigned long long *st = p.stats + (size_t)g * ST_N;
        st[ST_MOVES] += st_moves;
        st[ST_EVALS] += st_evals;
        st[ST_COMPACT] += st_compact;
        if (fault) atomicOr(p.faults, fault);
        if (queued) { // hand the copy to this launch's extra workgroups: everything they read (cj_*, which[g]) and the alloc
                      // placeholder they overwrite is stored above; the release makes it visible before the job is, and the job
                      // is acknowledged before this wave reports itself seen (az_advance_kernel)
            __hip_atomic_store(p.cj_job + row, g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The tick kernel.  Hot path: consume network results, then MCTS.playout until the network is needed again.
// MAPPED = false: the launch covers slots [g_first, g_end), row g of priors / values / obs_out belongs to slot g.
// MAPPED = true (az_engine_advance_rows, the thinned-out tail of a generation): it covers the first g_end entries of the
// dense list row_slot[]; entry i writes its request to row i, and reads the answer to its previous request from row req_row[g].
template <int GAME, int NP, bool MAPPED>
__device__ __forceinline__ void advance_slot_wave(const Params &p, const int g_first, const int g_end, const float *__restrict__ priors,
                                                  const float *__restrict__ values, float *__restrict__ obs_out) {
    const int lane = threadIdx.x & 63;
    const int row = g_first + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= g_end) return;
    const int g = MAPPED ? rfl(p.row_slot[row]) : row;
    float *__restrict__ const obs_row = obs_out + (size_t)row * p.obs_elems;
    // ---- 0. everything that is addressed by the slot index alone: ONE memory round trip ------------------------
    // (the kernel is a chain of dependent reads - PMC: waves parked in s_waitcnt 63 % of their life - so its duration
    //  is the number of round trips on the longest chain; see DESIGN.md section 3 and DESIGN_HISTORY.md section 3)
    const int ph_raw = p.phase[g];
    SlotRegs sr;
    const int lf_ply_raw = p.leaf_ply[g], lf_depth_raw = p.depth[g];
    const uint32_t lf_node_raw = p.leaf_node[g];
    const uint64_t lf_bb0_raw = p.leaf_bb0[g], lf_bb1_raw = p.leaf_bb1[g];
    uint32_t path_raw[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) path_raw[i] = p.path[(size_t)g * p.pstride + i * 64 + lane];
    const int ans_row = MAPPED ? rfl(p.req_row[g]) : g;
    const float value_raw = (values && !MAPPED) ? values[g] : 0.f;
    slot_load(p, g, sr);
    const int ph = rfl(ph_raw);
    if (ph == PH_MOVE || ph == PH_NEED_ROOT || ph == PH_OPP_DONE) { // the agent's move / the opponent's move / the next search's
                                                                    // root request: ends this slot's tick
        if (MAPPED && lane == 0) p.req_row[g] = row;
        move_step<GAME>(p, g, lane, ph, sr, obs_row, row);
        return;
    }
    if (ph != PH_RUN && ph != PH_WAIT_LEAF && ph != PH_WAIT_ROOT) return;

    const AzGeom &geom = p.geom;
    Pool t = pool_at(p, sr.pool);
    unsigned long long st_sims = 0, st_evals = 0, st_term = 0, st_depth = 0, st_children = 0, st_nodes = 0;
    unsigned int fault = 0;
    Path<NP> path;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        path.r[i] = 0;
        path.pn[i] = 0;
        path.pq[i] = 0.0;
    }

    // ---- 1. second round trip: the root node, and (request outstanding) the nodes of the recorded path ----------
    const AzNode rootn = t.nd[sr.root]; // same address in every lane
    uint32_t root_n = rflu(rootn.N), root_c0 = rflu(rootn.C0), root_meta = rflu(rootn.META);
    if (lane == 0) { // depth 0 of every path
        path.r[0] = sr.root;
        path.pn[0] = rootn.N;
        path.pq[0] = rootn.Q;
    }
    const float value_ans = MAPPED ? ((values && (ph == PH_WAIT_LEAF || ph == PH_WAIT_ROOT)) ? values[ans_row] : 0.f) : value_raw;
    if (ph == PH_WAIT_LEAF || ph == PH_WAIT_ROOT) {
        const float *pri = priors + (size_t)ans_row * p.A;
        AzState ls;
        uint32_t node, node_act;
        int depth = 0;
        bool fresh;
        uint32_t c0;
        if (ph == PH_WAIT_LEAF) {
            ls.bb0 = rfl64(lf_bb0_raw);
            ls.bb1 = rfl64(lf_bb1_raw);
            ls.ply = rfl(lf_ply_raw);
            node = rflu(lf_node_raw);
            const int dm = rfl(lf_depth_raw); // depth | action leading to the leaf << 16
            depth = dm & 0xFFFF;
            node_act = (uint32_t)dm >> 16;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                path.r[i] = path_raw[i];
                if (i * 64 + lane <= depth) { // N and Q of the path nodes, BEFORE anything of this tick is stored
                    path.pn[i] = t.nd[path_raw[i]].N;
                    path.pq[i] = t.nd[path_raw[i]].Q;
                }
            }
            fresh = true; // a request is only ever issued for an unexpanded node (nc == 0 below)
            c0 = sr.alloc;
        } else {
            ls = sr.rs;
            node = sr.root;
            node_act = root_meta & 0xFFFFu;
            c0 = root_c0;
            fresh = (c0 == NONE32);
            if (fresh) c0 = sr.alloc;
        }
        constexpr int MAXM = GAME == AZG_CONNECT_FOUR ? 1 : 3; // moves a lane can own (a column / a cell's three directions)
        int k[3], act[3], mine;
        int n = enum_moves<GAME>(ls, geom, lane, k, act, mine);
        if (fresh && sr.alloc + (uint32_t)n > p.cap) {
            fault |= AZ_FAULT_POOL_EXHAUSTED;
            n = 0;
            mine = 0;
        }
        double eta[3] = {0.0, 0.0, 0.0};
        if (ph == PH_WAIT_ROOT) { // mcts.py:182-190
            // injected draws (parity mode) or the Philox draws move_step staged with the request
            const double *e = p.rng_mode == AZ_RNG_INJECTED
                                  ? p.etas + ((size_t)sr.gid * p.max_plies + sr.rs.ply) * p.maxc
                                  : p.eta_buf + (size_t)g * p.maxc;
#pragma unroll
            for (int j = 0; j < MAXM; j++)
                if (j < mine) eta[j] = e[k[j]];
        }
        float pf[3] = {0.f, 0.f, 0.f}; // all the lane's priors first: independent loads, one round trip
#pragma unroll
        for (int j = 0; j < MAXM; j++)
            if (j < mine) pf[j] = pri[act[j]];
#pragma unroll
        for (int j = 0; j < MAXM; j++)
            if (j < mine) {
                if (!(pf[j] == pf[j])) fault |= AZ_FAULT_BAD_PRIOR;
                double pv = (double)pf[j];
                if (ph == PH_WAIT_ROOT) pv = p.one_minus_ratio * pv + 0.25 * eta[j]; // literal 0.25: mcts.py:189
                uint32_t i = c0 + (uint32_t)k[j];
                if (fresh) { // mcts.py:63-64: Node(parent, prior)
                    AzNode nn = {0u, NONE32, (uint32_t)act[j], 0u, 0.0, pv};
                    t.nd[i] = nn;
                } else {
                    t.nd[i].P = pv;
                }
            }
        if (fresh && n > 0) {
            if (lane == 0) {
                t.nd[node].C0 = c0;
                t.nd[node].META = node_act | ((uint32_t)n << 16);
            }
            if (node == sr.root) { // the root itself was the unexpanded node (root request, or a depth-0 leaf request)
                root_c0 = c0;
                root_meta = node_act | ((uint32_t)n << 16);
            }
            sr.alloc += (uint32_t)n;
            st_nodes += (unsigned long long)n;
        }
        if (ph == PH_WAIT_LEAF) { // mcts.py:152: node.update_recursive(-leaf_value)
            float vf = value_ans;
            if (!(vf == vf)) fault |= AZ_FAULT_BAD_PRIOR;
            backup_path<NP>(t, path, depth, -(double)vf, lane);
            root_n++;
            sr.sims++;
            st_sims++;
            st_depth += (unsigned long long)depth;
        }
        __threadfence_block(); // this wave's stores before its next reads of the same nodes
    }

    // ---- 2. playouts until the network is needed again -------------------------------------------
    int budget = p.max_sims_per_tick;
    int next_phase = PH_RUN;
    const unsigned long long t_start = wall_clock64(); // 100 MHz
    // The root's children, kept in registers across the playouts CHAINED in this launch (a chained playout follows a terminal
    // hit, which expands nothing and changes only N and Q of the nodes on its path): the launch is as long as its slowest wave,
    // and that wave is a chain - one HBM round trip less per chained playout.
    bool rc_valid = false;
    int rc_best = 0;
    uint32_t rc_n = 0, rc_c0 = NONE32, rc_meta = 0;
    double rc_q = 0.0, rc_p = 0.0;
    for (;;) {
        if (__ballot(fault != 0)) { // faults are raised per lane: make the exit wave-uniform
            next_phase = PH_IDLE;
            break;
        }
        if (sr.sims >= p.S) { // the agent's move step (move_step) opens this slot's next tick
            next_phase = p.manual_moves ? PH_SEARCH_DONE : PH_MOVE;
            break;
        }
        if (budget-- <= 0) {
            next_phase = PH_RUN;
            break;
        }
        // chained (NN-free) playouts only start early in the launch: the launch lasts as long as its slowest wave
        if (p.chain_clocks && budget + 1 < p.max_sims_per_tick && wall_clock64() - t_start > (unsigned long long)p.chain_clocks) {
            next_phase = PH_RUN;
            break;
        }
        // ================= MCTS.playout (mcts.py:126-153) =================
        // The root's N / first child / child count are carried in registers from the load above (N grows by one per
        // playout); lane 0 of the path carries its N and Q.
        AzState s = sr.rs;
        uint32_t node = sr.root;
        uint32_t np_ = root_n;
        uint32_t c0 = root_c0;
        uint32_t meta = root_meta;
        int nc = c0 == NONE32 ? 0 : (int)(meta >> 16);
        int depth = 0, term = 0;
        float ret0 = 0.f;
        int mover = s.ply & 1;
        while (nc > 0 && !term) {
            mover = s.ply & 1;
            double val = -INFINITY, cq = 0.0;
            uint32_t cn = 0, cc0 = NONE32, cmeta = 0;
            if (lane < nc) {
                AzNode c;
                if (depth == 0 && rc_valid) { // (wave-uniform) a chained playout: the root's children are still in registers
                    c.N = rc_n;
                    c.C0 = rc_c0;
                    c.META = rc_meta;
                    c.Q = rc_q;
                    c.P = rc_p;
                } else {
                    c = t.nd[c0 + lane]; // the lane's child: two 16-byte loads
                    if (depth == 0) {
                        rc_n = c.N;
                        rc_c0 = c.C0;
                        rc_meta = c.META;
                        rc_q = c.Q;
                        rc_p = c.P;
                    }
                }
                cn = c.N;
                cc0 = c.C0;
                cmeta = c.META;
                cq = c.Q;
                if (sr.rule == AZ_SELECT_PUCT)
                    val = c.Q + ((p.c_puct * c.P) * sqrt((double)np_)) / (double)(cn + 1); // mcts.py:78
                else // mcts.py:80; log(N_parent) from the host's table (N_parent >= 1 whenever a child has a visit)
                    val = cn == 0 ? INFINITY : c.Q + (p.c_puct * c.P) * sqrt(p.log_table[np_ < p.log_n ? np_ : 0u] / (double)cn);
            }
            double mx = wave_max(val);
            unsigned long long eq = __ballot(val == mx);
            int best = eq ? __ffsll(eq) - 1 : 0; // first maximum in child order (mcts.py:50)
            if (depth == 0) rc_best = best;
            if (!eq) fault |= AZ_FAULT_BAD_PRIOR;
            if (sr.rule != AZ_SELECT_PUCT && np_ >= p.log_n) fault |= AZ_FAULT_VISIT_RANGE;
            st_children += (unsigned long long)nc;
            node = c0 + (uint32_t)best;
            np_ = (uint32_t)__builtin_amdgcn_readlane((int)cn, best);
            c0 = (uint32_t)__builtin_amdgcn_readlane((int)cc0, best);
            meta = (uint32_t)__builtin_amdgcn_readlane((int)cmeta, best);
            const double qsel = readlane_d(cq, best);
            nc = c0 == NONE32 ? 0 : (int)(meta >> 16);
            term = az_apply<GAME>(s, geom, (int)(meta & 0xFFFFu), &ret0);
            depth++;
            path.set(lane, depth, node, np_, qsel);
        }
        if (term) { // mcts.py:148-152: leaf_value = -player_return(mover); update_recursive(-leaf_value)
            double x = mover == 0 ? (double)ret0 : -(double)ret0;
            backup_path<NP>(t, path, depth, x, lane);
            { // the backed-up N and Q of the depth-1 node (lane 1 of the path registers) into the cached child record
                const uint32_t n1 = (uint32_t)__builtin_amdgcn_readlane((int)path.pn[0], 1);
                const double q1 = readlane_d(path.pq[0], 1);
                if (lane == rc_best) {
                    rc_n = n1;
                    rc_q = q1;
                }
                rc_valid = true;
            }
            __threadfence_block();
            root_n++;
            sr.sims++;
            st_sims++;
            st_term++;
            st_depth += (unsigned long long)depth;
            continue;
        }
        // non-terminal leaf: ask the network (mcts.py:146)
        write_obs<GAME>(p, s, obs_row, lane);
        st_evals++;
        if (lane == 0) {
            if (MAPPED) p.req_row[g] = row;
            p.leaf_bb0[g] = s.bb0;
            p.leaf_bb1[g] = s.bb1;
            p.leaf_ply[g] = s.ply;
            p.leaf_node[g] = node;
            p.depth[g] = depth | (int)((meta & 0xFFFFu) << 16); // + the action leading to the leaf (its META low half)
        }
#pragma unroll
        for (int i = 0; i < NP; i++) p.path[(size_t)g * p.pstride + i * 64 + lane] = path.r[i];
        next_phase = PH_WAIT_LEAF;
        break;
    }

    // ---- 3. write the slot back ----------------------------------------------------------------
    fault = wave_or(fault);
    if (lane == 0) {
        slot_store(p, g, sr, next_phase);
        unsigned long long *st = p.stats + (size_t)g * ST_N;
        st[ST_SIMS] += st_sims;
        st[ST_EVALS] += st_evals;
        st[ST_TERM] += st_term;
        st[ST_DEPTH] += st_depth;
        st[ST_CHILDREN] += st_children;
        st[ST_NODES] += st_nodes;
        if (fault) atomicOr(p.faults, fault);
    }
}

template <int GAME, int NP, bool MAPPED>
__global__ __launch_bounds__(256) void az_advance_kernel(Params p, const int g_first, const int g_end, const float *__restrict__ priors,
                                                         const float *__restrict__ values, float *__restrict__ obs_out) {
    const int n_slot_wgs = (g_end - g_first + 3) >> 2;
    if ((int)blockIdx.x >= n_slot_wgs) { // (only launched when p.defer_compact) the copies this launch's slot waves hand over
        compact_jobs(p, (int)blockIdx.x - n_slot_wgs, (int)gridDim.x - n_slot_wgs, g_end - g_first);
        return;
    }
    const int epoch = p.defer_compact ? p.cjob_count[0] : 0; // (the same in every wave of the launch: compact_jobs)
    advance_slot_wave<GAME, NP, MAPPED>(p, g_first, g_end, priors, values, obs_out);
    // every slot wave reports itself seen, whichever way it left: its extra workgroup stays until none of its rows can hand over a job
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p.defer_compact && (threadIdx.x & 63) == 0 && g_first + row < g_end)
        __hip_atomic_store(p.cj_seen + row, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// Arena opponents (evaluation games, game_utils.py:16-145): ONE THREAD per slot whose opponent is to move.
//  * AZ_OPPONENT_RANDOM - pyspiel.make_uniform_random_bot: a uniformly random legal action.
//  * AZ_OPPONENT_UCT    - open_spiel.python.algorithms.mcts.MCTSBot(game, player, uct_c, max_search_nodes,
//    RandomRolloutEvaluator(1)) with its defaults solve=True, child_selection_fn=uct_value.  OpenSpiel is a third-party
//    dependency absent from the reference tree and unpinned; this restates the published algorithm: per
//    simulation descend from the root while the node has been visited (a visited node gets its children the first time it is
//    descended through, in an order SHUFFLED by the bot's random state - here Fisher-Yates on the Philox stream), choosing
//    the child that maximises its proven outcome for the mover if it has one, else  total_reward / explore_count + uct_c *
//    sqrt(log(parent explore_count) / explore_count)  (an unvisited child counts as +infinity; first maximum in child
//    order); a terminal node takes its returns as outcome, any other leaf ONE uniformly random rollout to the end of the
//    game; the result is added to every node of the path as seen by the player who moved into it; MCTS-Solver: while the
//    backup is "solved", a node all of whose children are solved, or one of whose children is a proven win for the player
//    to move, takes the outcome of its best child; the search stops when the root is solved; finally the child with the
//    largest (proven outcome for the mover or 0, explore_count, total_reward) is played (first maximum).
// Random numbers: Philox stream (seed, game id, ply, purpose 2), consumed in simulation order, so that the C restatement
// in oracle/az_oracle.c reproduces every move bit for bit.  META: action | n_children << 16 | outcome code << 24
// (0 = open, 1 / 2 / 3 = player 0's return is -1 / 0 / +1).
#define UCT_NCH(m) (int)(((m) >> 16) & 0xFFu)
#define UCT_OUT(m) (int)(((m) >> 24) & 0x7u)
template <int GAME> __device__ int uct_search(const Params &p, int g, const AzState &root_s, Philox &r) {
    const AzGeom &geom = p.geom;
    uint32_t *N = p.uct_N + (size_t)g * p.uct_cap, *C0 = p.uct_C0 + (size_t)g * p.uct_cap, *META = p.uct_META + (size_t)g * p.uct_cap;
    double *W = p.uct_W + (size_t)g * p.uct_cap;
    uint32_t alloc = 1;
    N[0] = 0;
    W[0] = 0.0;
    C0[0] = NONE32;
    META[0] = 0;
    const int root_player = root_s.ply & 1;
    uint32_t path[192];
    for (int sim = 0; sim < p.opp_sims; sim++) {
        AzState s = root_s;
        uint32_t node = 0;
        int term = 0, depth = 0;
        float ret0 = 0.f;
        path[0] = 0;
        while (!term && N[node] > 0) {
            const int to_move = (root_player + depth) & 1; // the player who moves INTO the children
            if (C0[node] == NONE32) { // first descent through a visited node: create its children, shuffled
                const int n = az_count_legal<GAME>(s, geom);
                if (alloc + (uint32_t)n > p.uct_cap) return -1;
                C0[node] = alloc;
                META[node] = (META[node] & 0xFF00FFFFu) | ((uint32_t)n << 16);
                for (int k = 0; k < n; k++) {
                    N[alloc + k] = 0;
                    W[alloc + k] = 0.0;
                    C0[alloc + k] = NONE32;
                    META[alloc + k] = (uint32_t)az_nth_legal<GAME>(s, geom, k);
                }
                for (int i = n - 1; i >= 1; i--) { // random_state.shuffle
                    int j = (int)(philox_u01(r) * (double)(i + 1));
                    j = j > i ? i : j;
                    const uint32_t t = META[alloc + i];
                    META[alloc + i] = META[alloc + j];
                    META[alloc + j] = t;
                }
                alloc += (uint32_t)n;
            }
            const uint32_t c0 = C0[node];
            const int nc = UCT_NCH(META[node]);
            const double L = p.log_table[N[node]];
            double best = -INFINITY;
            int bi = 0;
            for (int k = 0; k < nc; k++) {
                const uint32_t cn = N[c0 + k];
                const int oc = UCT_OUT(META[c0 + k]);
                const double v = oc ? (to_move == 0 ? (double)(oc - 2) : -(double)(oc - 2))
                                    : (cn == 0 ? INFINITY : W[c0 + k] / (double)cn + p.opp_c * sqrt(L / (double)cn));
                if (v > best) {
                    best = v;
                    bi = k;
                }
            }
            node = c0 + (uint32_t)bi;
            term = az_apply<GAME>(s, geom, (int)(META[node] & 0xFFFFu), &ret0);
            path[++depth] = node;
        }
        bool solved = false;
        if (term) { // a terminal node takes its returns as outcome
            META[node] = (META[node] & 0xF8FFFFFFu) | ((uint32_t)((int)ret0 + 2) << 24);
            solved = true;
        }
        while (!term) { // RandomRolloutEvaluator(1): uniformly random legal actions to the end of the game
            const int n = az_count_legal<GAME>(s, geom);
            int k = (int)(philox_u01(r) * (double)n);
            k = k < n ? k : n - 1;
            term = az_apply<GAME>(s, geom, az_nth_legal<GAME>(s, geom, k), &ret0);
        }
        for (int d = depth; d >= 0; d--) { // the node at depth d >= 1 was entered by player (root_player + d + 1) & 1; the root
                                           // carries the player to move
            const int mover = d == 0 ? root_player : (root_player + d + 1) & 1;
            const uint32_t nd = path[d];
            N[nd] += 1;
            W[nd] += mover == 0 ? (double)ret0 : -(double)ret0;
            if (solved && C0[nd] != NONE32) { // MCTS-Solver backup
                const int player = (root_player + d) & 1; // the player to move at nd
                const uint32_t c0 = C0[nd];
                const int nc = UCT_NCH(META[nd]);
                int best_oc = 0, best_v = -2;
                bool all_solved = true;
                for (int k = 0; k < nc; k++) {
                    const int oc = UCT_OUT(META[c0 + k]);
                    if (!oc) all_solved = false;
                    else {
                        const int vc = player == 0 ? oc - 2 : 2 - oc;
                        if (vc > best_v) {
                            best_v = vc;
                            best_oc = oc;
                        }
                    }
                }
                if (best_oc && (all_solved || best_v == 1)) META[nd] = (META[nd] & 0xF8FFFFFFu) | ((uint32_t)best_oc << 24);
                else solved = false;
            }
        }
        if (UCT_OUT(META[0])) break; // the root is solved
    }
    if (C0[0] == NONE32) return -1;
    const uint32_t c0 = C0[0];
    const int nc = UCT_NCH(META[0]);
    int bi = -1, bo = 0;
    for (int k = 0; k < nc; k++) { // largest (proven outcome for the mover or 0, explore_count, total_reward); first maximum
        const int oc = UCT_OUT(META[c0 + k]);
        const int o = oc ? (root_player == 0 ? oc - 2 : 2 - oc) : 0;
        if (bi < 0 || o > bo || (o == bo && (N[c0 + k] > N[c0 + bi] || (N[c0 + k] == N[c0 + bi] && W[c0 + k] > W[c0 + bi])))) {
            bi = k;
            bo = o;
        }
    }
    return (int)(META[c0 + bi] & 0xFFFFu);
}

template <int GAME> __global__ __launch_bounds__(64) void az_opponent_kernel(Params p) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.G || p.phase[g] != PH_OPPONENT || p.opp_kind == AZ_OPPONENT_EXTERNAL) return;
    AzState s;
    s.bb0 = p.bb0[g];
    s.bb1 = p.bb1[g];
    s.ply = p.ply[g];
    Philox r;
    philox_init(r, p.seed, (uint32_t)p.gid[g], (uint32_t)s.ply, 2u, 0u);
    int action;
    if (p.opp_kind == AZ_OPPONENT_RANDOM) {
        const int n = az_count_legal<GAME>(s, p.geom);
        int k = (int)(philox_u01(r) * (double)n);
        action = az_nth_legal<GAME>(s, p.geom, k < n ? k : n - 1);
    } else {
        action = uct_search<GAME>(p, g, s, r);
    }
    if (action < 0) {
        atomicOr(p.faults, AZ_FAULT_POOL_EXHAUSTED);
        p.phase[g] = PH_IDLE;
        return;
    }
    p.opp_action[g] = action;
    p.phase[g] = PH_OPP_DONE;
}

// MCTS.update_root for manual_moves engines (AlphaZeroBot.step outside the self-play loop).
template <int GAME>
__global__ __launch_bounds__(256) void az_update_root_kernel(Params p, const int *__restrict__ actions, int keep_subtree) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= p.G) return;
    int action = rfl(actions[g]);
    int ph = rfl(p.phase[g]);
    if (action == AZ_ACTION_SEARCH_AGAIN) { // MCTS.search() called again at an unchanged root (mcts.py:164-180): another
                                            // S playouts on the same tree, root re-expanded with a fresh Dirichlet draw
        if (ph == PH_SEARCH_DONE && lane == 0) {
            p.sims[g] = 0;
            p.phase[g] = p.use_dirichlet ? PH_NEED_ROOT : PH_RUN;
        }
        return;
    }
    if (action < 0 || ph == PH_IDLE) return;
    SlotRegs sr;
    slot_load(p, g, sr);
    Pool t = pool_at(p, sr.pool);
    unsigned int fault = 0;
    unsigned long long st_compact = 0;
    { // the action must be legal in the root state (apply_action on an illegal move would corrupt the bitboards)
        int k[3], act[3], mine;
        enum_moves<GAME>(sr.rs, p.geom, lane, k, act, mine);
        bool is_it = false;
#pragma unroll
        for (int j = 0; j < 3; j++) is_it |= j < mine && act[j] == action;
        if (!__ballot(is_it)) {
            if (lane == 0) {
                p.phase[g] = PH_IDLE;
                atomicOr(p.faults, AZ_FAULT_ILLEGAL_ACTION);
            }
            return;
        }
    }
    uint32_t c0 = rflu(t.nd[sr.root].C0);
    int nc = c0 == NONE32 ? 0 : (int)(rflu(t.nd[sr.root].META) >> 16);
    int cact = lane < nc ? (int)(t.nd[c0 + lane].META & 0xFFFFu) : -1;
    unsigned long long hit = __ballot(cact == action);
    int sel = (keep_subtree && hit) ? __ffsll(hit) - 1 : -1;
    float ret0 = 0.f;
    int term = az_apply<GAME>(sr.rs, p.geom, action, &ret0);
    sr.sims = 0;
    reroot(p, g, sr, t, sel, !keep_subtree, keep_subtree ? p.select_rule : AZ_SELECT_PUCT, lane, fault, st_compact);
    fault = wave_or(fault);
    if (lane == 0) {
        slot_store(p, g, sr, (term || fault) ? PH_IDLE : (p.use_dirichlet ? PH_NEED_ROOT : PH_RUN));
        p.stats[(size_t)g * ST_N + ST_COMPACT] += st_compact;
        if (fault) atomicOr(p.faults, fault);
    }
}

__global__ void az_reset_kernel(Params p) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.G) return;
    bool active = (long long)g < p.n_games;
    p.phase[g] = !active ? PH_IDLE
                 : ((p.arena_agent != AZ_ARENA_SELF_PLAY && ((p.start.ply ^ g ^ p.arena_flip) & 1)) ? PH_OPPONENT
                                                                                       : (p.use_dirichlet ? PH_NEED_ROOT : PH_RUN));
    p.gid[g] = active ? g : -1;
    p.bb0[g] = p.start.bb0;
    p.bb1[g] = p.start.bb1;
    p.ply[g] = p.start.ply;
    p.sims[g] = 0;
    p.which[g] = g | (!p.keep_tree || p.manual_moves ? AZ_SELECT_PUCT
                      : (p.start.ply >= (p.arena_agent == AZ_ARENA_SELF_PLAY ? 1 : 2) ? p.select_rule : AZ_SELECT_PUCT)) << 30; // pool g, start_rule
    for (int k = g; k < p.n_spare; k += p.G) p.spare[k] = p.G + k; // the spare pools follow the slots' own
    if (g == 0) p.cjob_count[0] = 1, p.cjob_count[1] = p.cjob_count[2] = 0; // epoch 1: no row has been seen in it
    p.cj_job[g] = 0;
    p.cj_seen[g] = 0;
    p.root[g] = 0;
    p.alloc[g] = 1;
    p.depth[g] = 0;
    p.leaf_node[g] = 0;
    size_t base = (size_t)g * p.cap;
    AzNode root0 = {0u, NONE32, 0u, 0u, 0.0, 0.0};
    p.nodes[base] = root0;
    for (int i = 0; i < ST_N; i++) p.stats[(size_t)g * ST_N + i] = 0;
    if (g == 0) {
        *p.next_game = (unsigned long long)(p.n_games < p.G ? p.n_games : p.G);
        *p.games_done = 0;
        *p.faults = 0;
    }
}

// ================================================================================================
// host side: C ABI
static std::string g_create_err;

#define HIPCHK(e, call)                                                                      \
    do {                                                                                     \
        hipError_t _s = (call);                                                              \
        if (_s != hipSuccess) {                                                              \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_s);                    \
            return AZ_E_HIP;                                                                 \
        }                                                                                    \
    } while (0)

template <typename T> static int dalloc(az_engine *e, T **out, size_t count) {
    void *ptr = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t s = hipMalloc(&ptr, bytes);
    if (s != hipSuccess) {
        e->err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(s);
        return AZ_E_NOMEM;
    }
    e->dev_allocs.push_back(ptr);
    e->sizes.device_bytes += (int64_t)bytes;
    *out = (T *)ptr;
    return AZ_OK;
}

static void pw_build(PwPlan &pw, int lo, int n) { // numpy pairwise_sum recursion (PW_BLOCKSIZE 128)
    if (n <= 128) {
        pw.lo[pw.n_blocks] = lo;
        pw.len[pw.n_blocks] = n;
        pw.ops[pw.n_ops++] = pw.n_blocks++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    pw_build(pw, lo, n2);
    pw_build(pw, lo + n2, n - n2);
    pw.ops[pw.n_ops++] = -1;
}

extern "C" const char *az_last_error(const az_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

extern "C" int az_engine_destroy(az_engine *e) {
    if (!e) return AZ_OK;
    (void)hipSetDevice(e->cfg.device);
    for (void *ptr : e->dev_allocs) (void)hipFree(ptr);
    if (e->d_etas) (void)hipFree(e->d_etas);
    if (e->d_us) (void)hipFree(e->d_us);
    if (e->d_actions) (void)hipFree(e->d_actions);
    delete e;
    return AZ_OK;
}

extern "C" int az_engine_create(const az_config *cfg, az_engine **out) {
    if (!cfg || !out) {
        g_create_err = "null argument";
        return AZ_E_INVALID;
    }
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(az_config)) {
        g_create_err = "az_config.struct_size mismatch (header/library skew)";
        return AZ_E_INVALID;
    }
    az_config c = *cfg;
    if (c.game == AZ_GAME_CONNECT_FOUR) {
        c.rows = 6;
        c.cols = 7;
    } else if (c.game == AZ_GAME_BREAKTHROUGH) {
        if (c.rows < 4 || c.cols < 2 || c.rows * c.cols > 64 || 6 * c.cols > 64) {
            g_create_err = "breakthrough board must satisfy rows>=4, rows*cols<=64, cols<=10";
            return AZ_E_INVALID;
        }
    } else {
        g_create_err = "unknown game id";
        return AZ_E_INVALID;
    }
    if (c.n_slots < 1 || c.n_playouts < 1 || c.max_games < 1 || !(c.temperature > 0.0) || !(c.c_puct >= 0.0)) {
        g_create_err = "n_slots, n_playouts, max_games must be >= 1; temperature > 0; c_puct >= 0";
        return AZ_E_INVALID;
    }
    if (c.arena_agent < AZ_ARENA_SELF_PLAY || c.arena_agent > AZ_ARENA_NET ||
        (c.arena_agent != AZ_ARENA_SELF_PLAY && (c.arena_opponent < AZ_OPPONENT_RANDOM || c.arena_opponent > AZ_OPPONENT_EXTERNAL)) ||
        (c.arena_flip != 0 && c.arena_flip != 1) || (c.arena_agent == AZ_ARENA_SELF_PLAY && c.arena_flip) ||
        (c.arena_agent == AZ_ARENA_SELF_PLAY && c.arena_opponent != AZ_OPPONENT_NONE) ||
        (c.arena_opponent == AZ_OPPONENT_UCT && (c.opponent_sims < 2 || c.opponent_sims > 100000 || !(c.opponent_uct_c >= 0.0))) ||
        (c.arena_agent != AZ_ARENA_SELF_PLAY && (c.manual_moves || c.rng_mode != AZ_RNG_PHILOX))) {
        g_create_err = "bad arena configuration (agent 0..2; an arena needs opponent RANDOM, UCT with 2 <= opponent_sims, or EXTERNAL; "
                       "rng_mode PHILOX, manual_moves 0)";
        return AZ_E_INVALID;
    }
    if ((c.select_rule != AZ_SELECT_PUCT && c.select_rule != AZ_SELECT_UCT) || c.spare_pools < 0 ||
        (c.arena_probabilistic != 0 && c.arena_probabilistic != 1) ||
        (c.arena_probabilistic && c.arena_agent != AZ_ARENA_ZERO)) {
        g_create_err = "select_rule must be AZ_SELECT_PUCT or AZ_SELECT_UCT; arena_probabilistic 0/1 (AZ_ARENA_ZERO only); "
                       "spare_pools >= 0";
        return AZ_E_INVALID;
    }
    if (!c.use_dirichlet && c.n_playouts < 2 && c.arena_agent != AZ_ARENA_NET) {
        g_create_err = "n_playouts must be >= 2 without root Dirichlet expansion (mcts.py:162 divides by zero)";
        return AZ_E_INVALID;
    }
    if (c.backup < 0 || c.backup > 3 || (c.rng_mode != AZ_RNG_PHILOX && c.rng_mode != AZ_RNG_INJECTED)) {
        g_create_err = "bad backup / rng_mode";
        return AZ_E_INVALID;
    }
    az_engine *e = new az_engine();
    e->cfg = c;
    memset(&e->p, 0, sizeof e->p);
    memset(&e->sizes, 0, sizeof e->sizes);
    hipError_t s = hipSetDevice(c.device);
    if (s != hipSuccess) {
        g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(s);
        delete e;
        return AZ_E_HIP;
    }
    Params &p = e->p;
    p.geom = az_make_geom(c.game, c.rows, c.cols);
    p.game = c.game;
    p.A = az_num_actions(c.game, c.rows, c.cols);
    p.maxc = az_max_children(c.game, c.rows, c.cols);
    p.max_plies = az_max_plies(c.game, c.rows, c.cols);
    p.obs_elems = 4 * c.rows * c.cols;
    p.pstride = c.game == AZ_GAME_CONNECT_FOUR ? 64 : 192;
    if (p.max_plies > p.pstride - 1) { // a select path holds one node per ply (Path<NP>: NP * 64 depths)
        g_create_err = "board too large: a game may last " + std::to_string(p.max_plies) + " plies, the select path holds " +
                       std::to_string(p.pstride - 1);
        delete e;
        return AZ_E_INVALID;
    }
    p.G = c.n_slots;
    p.S = c.n_playouts;
    p.use_dirichlet = c.use_dirichlet ? 1 : 0;
    p.keep_tree = c.keep_search_tree ? 1 : 0;
    p.backup = c.backup;
    p.rng_mode = c.rng_mode;
    // NN-free (terminal-leaf) playouts a slot may chain in one tick: a count cap, and a time window - a new one only
    // starts within the first chain_window_us of the launch (wall_clock64 ticks at 100 MHz).  Measured, connect_four
    // S=400 4096 slots: count cap 3 alone 2600 games/s; cap 6 + 10 us window 2700 (the launch lasts as long as its
    // slowest wave, ~7.5 us per chained playout; scheduling only - the games do not depend on it).
    p.max_sims_per_tick = c.max_sims_per_tick > 0 ? c.max_sims_per_tick : 10;
    p.chain_clocks = c.chain_window_us < 0 ? 0 : (c.chain_window_us > 0 ? c.chain_window_us : 10) * 100;
    p.manual_moves = c.manual_moves ? 1 : 0;
    p.arena_agent = c.arena_agent;
    p.opp_kind = c.arena_opponent;
    p.opp_sims = c.opponent_sims;
    p.arena_flip = c.arena_flip;
    p.opp_c = c.opponent_uct_c;
    p.uct_cap = c.arena_opponent == AZ_OPPONENT_UCT ? (uint32_t)(1 + (size_t)c.opponent_sims * p.maxc) : 0;
    p.need_per_move = (uint32_t)((c.n_playouts + 1) * p.maxc);
    // default pool: room for 48 searches, or for the whole game if it is shorter (connect_four: 42 plies -> a slot never
    // compacts), capped so that the pools take at most a quarter of the free HBM.  Measured, connect_four S=400, 4096 slots: 6 searches 10k compactions per
    // 4096 games, 24 searches 820, whole game none: +1.7 % games/s over 24, 31 GB instead of 18 GB.
    // Spare pools (round 3): a slot owns ONE pool; the target of a compaction comes from a shared set of spare pools (default
    // n_slots / 16, at least 16, at most n_slots) instead of a private second half per slot - the same capacity per slot in
    // half the memory (connect_four S=400, 4096 slots: 31 -> 16.5 GB; breakthrough 6x6 S=800: 155 -> 82 GB).
    p.n_spare = c.spare_pools > 0 ? c.spare_pools : (c.n_slots / 16 > 16 ? c.n_slots / 16 : 16);
    if (p.n_spare > c.n_slots) p.n_spare = c.n_slots;
    int64_t moves_room = p.max_plies < 48 ? p.max_plies : 48;
    int64_t cap = c.nodes_per_slot > 0 ? c.nodes_per_slot : moves_room * p.need_per_move + 64;
    if (c.nodes_per_slot <= 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // (a QUARTER of the free HBM - it was half until compaction left the critical path: breakthrough 6x6, S = 800, 4096
            //  slots: 155 GB and no compaction -> 78 GB, a few hundred cheap ones per 8192 games, -1..3 % games/s)
            int64_t fit = (int64_t)(free_b / 4 / ((size_t)(c.n_slots + p.n_spare) * sizeof(AzNode)));
            int64_t floor_cap = (int64_t)3 * p.need_per_move + 64;
            if (cap > fit) cap = fit > floor_cap ? fit : floor_cap;
        }
    }
    if (cap < (int64_t)p.need_per_move + 2 || cap > 0x7FFFFFFFll) {
        g_create_err = "nodes_per_slot too small for one search ((n_playouts+1)*max_children+2) or too large";
        delete e;
        return AZ_E_INVALID;
    }
    p.cap = (uint32_t)cap;
    p.defer_compact = 0;
    // (with tree reuse a pool grows by at most need_per_move nodes per move: one that holds max_plies searches never compacts)
    e->may_compact = c.keep_search_tree && !c.manual_moves && cap < (int64_t)p.max_plies * p.need_per_move + 64;
    p.c_puct = c.c_puct;
    p.one_minus_ratio = 1.0 - c.dirichlet_ratio;
    p.alpha = c.dirichlet_alpha > 0 ? c.dirichlet_alpha : 0.3;
    p.inv_temp = 1.0 / c.temperature;
    p.seed = c.seed;
    p.max_games = c.max_games;
    if (c.game == AZ_GAME_CONNECT_FOUR) az_init_state<AZG_CONNECT_FOUR>(p.start, p.geom);
    else az_init_state<AZG_BREAKTHROUGH>(p.start, p.geom);
    pw_build(p.pw, 0, p.A);

    az_sizes &z = e->sizes;
    z.num_actions = p.A;
    z.obs_planes = 4;
    z.rows = c.rows;
    z.cols = c.cols;
    z.max_children = p.maxc;
    z.max_plies = p.max_plies;
    z.n_slots = p.G;
    z.nodes_per_slot = cap;
    z.spare_pools = p.n_spare;
    z.max_games = c.max_games;

    size_t nodes = ((size_t)p.G + p.n_spare) * p.cap, G = (size_t)p.G;
    size_t plies = (size_t)c.max_games * p.max_plies;
    int rc = AZ_OK;
#define DA(ptr, n) if (rc == AZ_OK) rc = dalloc(e, &(ptr), (n))
    DA(p.nodes, nodes);
    DA(p.spare, (size_t)p.n_spare);
    DA(p.cj_job, G); DA(p.cj_seen, G); DA(p.cjob_count, 3); DA(p.cj_from, G); DA(p.cj_entry, G); DA(p.cj_root, G);
    DA(p.row_slot, G); DA(p.req_row, G); DA(p.n_rows_live, 1);
    DA(p.phase, G); DA(p.gid, G); DA(p.ply, G); DA(p.sims, G); DA(p.which, G); DA(p.depth, G); DA(p.leaf_ply, G);
    DA(p.root, G); DA(p.alloc, G); DA(p.leaf_node, G); DA(p.path, G * p.pstride);
    DA(p.bb0, G); DA(p.bb1, G); DA(p.leaf_bb0, G); DA(p.leaf_bb1, G);
    DA(p.stats, G * ST_N); DA(p.eta_buf, G * p.maxc);
    DA(p.next_game, 1); DA(p.games_done, 1); DA(p.faults, 1);
    if (c.arena_agent != AZ_ARENA_SELF_PLAY) DA(p.opp_action, G);
    double *d_log = nullptr;
    // log(n) table: the UCT opponent's explore counts stay <= opponent_sims; an AZ_SELECT_UCT tree's root gains n_playouts
    // visits per search, one search per ply (+ slack for MCTS.search called again at the same root)
    size_t log_n = 0;
    if (p.uct_cap) {
        DA(p.uct_N, G * p.uct_cap); DA(p.uct_C0, G * p.uct_cap); DA(p.uct_META, G * p.uct_cap); DA(p.uct_W, G * p.uct_cap);
        log_n = (size_t)c.opponent_sims + 2;
    }
    if (c.select_rule == AZ_SELECT_UCT) {
        size_t need = (size_t)c.n_playouts * ((size_t)p.max_plies + 2) + 2;
        if (need > log_n) log_n = need;
    }
    if (log_n) DA(d_log, log_n);
    p.log_n = (uint32_t)log_n;
    p.select_rule = c.select_rule;
    p.arena_prob = c.arena_probabilistic;
    p.n_prob_plies = c.num_probabilistic_actions > 0 ? c.num_probabilistic_actions : (c.num_probabilistic_actions < 0 ? 0 : 1000); // alphazerobot.py:36
    DA(p.rec_len, (size_t)c.max_games); DA(p.rec_ret0, (size_t)c.max_games);
    DA(p.rec_states, plies * 2); DA(p.rec_move, plies); DA(p.rec_nchild, plies);
    DA(p.rec_child_action, plies * p.maxc); DA(p.rec_child_visits, plies * p.maxc); DA(p.rec_value, plies);
#undef DA
    if (rc != AZ_OK) {
        g_create_err = e->err;
        az_engine_destroy(e);
        return rc;
    }
    (void)hipMemset(p.phase, 0, G * sizeof(int));
    (void)hipMemset(p.path, 0, G * p.pstride * sizeof(uint32_t));
    if (d_log) { // log(n) from the host's libm: the UCT values then agree bit for bit with the CPU restatement
        std::vector<double> lt(log_n, 0.0);
        for (size_t i = 1; i < lt.size(); i++) lt[i] = log((double)i);
        (void)hipMemcpy(d_log, lt.data(), lt.size() * sizeof(double), hipMemcpyHostToDevice);
        p.log_table = d_log;
    }
    *out = e;
    return AZ_OK;
}

extern "C" int az_engine_sizes(const az_engine *e, az_sizes *out) {
    if (!e || !out) return AZ_E_INVALID;
    *out = e->sizes;
    return AZ_OK;
}

extern "C" int az_engine_reset(az_engine *e, uint64_t seed, int64_t n_games, void *stream) {
    if (!e) return AZ_E_INVALID;
    if (n_games < 1 || n_games > e->cfg.max_games) {
        e->err = "n_games must be in [1, max_games]";
        return AZ_E_INVALID;
    }
    HIPCHK(e, hipSetDevice(e->cfg.device));
    e->p.seed = seed;
    e->p.n_games = n_games;
    e->n_games = n_games;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipMemsetAsync(e->p.rec_len, 0, sizeof(int) * (size_t)e->cfg.max_games, st));
    hipLaunchKernelGGL(az_reset_kernel, dim3((e->p.G + 255) / 256), dim3(256), 0, st, e->p);
    HIPCHK(e, hipGetLastError());
    e->reset_done = true;
    e->ticks = 0;
    e->rows_mapped = false;
    e->rows_live = 0;
    return AZ_OK;
}

extern "C" int az_engine_set_injected_rng(az_engine *e, const double *etas, const double *us, int64_t n_games) {
    if (!e || !etas || !us || n_games < 1) return AZ_E_INVALID;
    if (e->cfg.rng_mode != AZ_RNG_INJECTED) {
        e->err = "engine was not created with rng_mode = AZ_RNG_INJECTED";
        return AZ_E_STATE;
    }
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (e->d_etas) (void)hipFree(e->d_etas);
    if (e->d_us) (void)hipFree(e->d_us);
    e->d_etas = e->d_us = nullptr;
    size_t ne = (size_t)n_games * e->p.max_plies * e->p.maxc, nu = (size_t)n_games * e->p.max_plies;
    HIPCHK(e, hipMalloc((void **)&e->d_etas, ne * sizeof(double)));
    HIPCHK(e, hipMalloc((void **)&e->d_us, nu * sizeof(double)));
    HIPCHK(e, hipMemcpy(e->d_etas, etas, ne * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->d_us, us, nu * sizeof(double), hipMemcpyHostToDevice));
    e->p.etas = e->d_etas;
    e->p.us = e->d_us;
    e->inj_games = n_games;
    return AZ_OK;
}

extern "C" int az_engine_set_start_prefix(az_engine *e, const int32_t *actions, int32_t n) {
    if (!e || (n > 0 && !actions)) return AZ_E_INVALID;
    AzState s;
    float ret0 = 0.f;
    int term = 0;
    if (e->cfg.game == AZ_GAME_CONNECT_FOUR) {
        az_init_state<AZG_CONNECT_FOUR>(s, e->p.geom);
        for (int i = 0; i < n && !term; i++) {
            if (actions[i] < 0 || actions[i] > 6 || !((az_c4_legal_mask(s) >> actions[i]) & 1u)) {
                e->err = "illegal prefix action";
                return AZ_E_INVALID;
            }
            term = az_apply<AZG_CONNECT_FOUR>(s, e->p.geom, actions[i], &ret0);
        }
    } else {
        az_init_state<AZG_BREAKTHROUGH>(s, e->p.geom);
        for (int i = 0; i < n && !term; i++) {
            int a = actions[i];
            if (a < 0 || a >= e->p.A) {
                e->err = "illegal prefix action";
                return AZ_E_INVALID;
            }
            int d = (a >> 1) % 6, cell = (a >> 1) / 6, me = s.ply & 1;
            uint32_t mv = az_bt_cell_moves(s, e->p.geom, cell);
            int dd = d - (me ? 3 : 0);
            if (dd < 0 || dd > 2 || !(mv & (1u << dd)) || (int)((mv >> (4 + dd)) & 1u) != (a & 1)) {
                e->err = "illegal prefix action";
                return AZ_E_INVALID;
            }
            term = az_apply<AZG_BREAKTHROUGH>(s, e->p.geom, a, &ret0);
        }
    }
    if (term) {
        e->err = "prefix ends the game";
        return AZ_E_INVALID;
    }
    e->p.start = s;
    e->reset_done = false; // caller must reset again so that slots pick the new start up
    return AZ_OK;
}

// `defer`: compactions of this launch are handed to its own extra workgroups (compact_jobs; whole-engine launches only: the job
// list and its counters are one set per engine, and slot groups ticking on their own streams would share them).
template <bool MAPPED>
static int advance_range(az_engine *e, int g_first, int g_end, const float *priors, const float *values, float *obs_out, void *stream,
                         bool defer) {
    HIPCHK(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    defer = defer && e->may_compact;
    dim3 grid((g_end - g_first + 3) / 4 + (defer ? AZ_COMPACT_WGS : 0)), block(256);
    e->p.defer_compact = defer ? 1 : 0;
    if (e->cfg.game == AZ_GAME_CONNECT_FOUR) {
        hipLaunchKernelGGL((az_advance_kernel<AZG_CONNECT_FOUR, 1, MAPPED>), grid, block, 0, st, e->p, g_first, g_end, priors, values, obs_out);
    } else {
        hipLaunchKernelGGL((az_advance_kernel<AZG_BREAKTHROUGH, 3, MAPPED>), grid, block, 0, st, e->p, g_first, g_end, priors, values, obs_out);
    }
    e->p.defer_compact = 0;
    HIPCHK(e, hipGetLastError());
    return AZ_OK;
}

// The slots that still play, in slot order, as a dense list (one workgroup: a block-wide prefix sum over the phase flags).
__global__ __launch_bounds__(1024) void az_compact_rows_kernel(Params p, int first_time) {
    __shared__ int wave_tot[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int g0 = 0; g0 < p.G; g0 += 1024) {
        const int g = g0 + tid;
        const bool live = g < p.G && p.phase[g] != PH_IDLE;
        if (first_time && g < p.G) p.req_row[g] = g; // until now row g held slot g's request
        const unsigned long long m = __ballot(live);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; w++) off += wave_tot[w];
        if (live) p.row_slot[off + before] = g;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += wave_tot[w];
            base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) *p.n_rows_live = base;
}

static int advance_checks(az_engine *e, const float *obs_out) {
    if (!e || !obs_out) return AZ_E_INVALID;
    if (!e->reset_done) {
        e->err = "az_engine_advance before az_engine_reset";
        return AZ_E_STATE;
    }
    if (e->cfg.rng_mode == AZ_RNG_INJECTED && (!e->p.etas || e->inj_games < e->n_games)) {
        e->err = "rng_mode INJECTED but az_engine_set_injected_rng was not called for all games";
        return AZ_E_STATE;
    }
    return AZ_OK;
}

extern "C" int az_engine_advance(az_engine *e, const float *priors, const float *values, float *obs_out, void *stream) {
    int rc = advance_checks(e, obs_out);
    if (rc != AZ_OK) return rc;
    bool pending = e->ticks > 0; // slots can only be waiting for the network after a first tick
    if (pending && (!priors || !values)) {
        e->err = "az_engine_advance: priors/values may be NULL only on the first tick after reset";
        return AZ_E_INVALID;
    }
    if (e->rows_mapped) {
        e->err = "az_engine_advance after az_engine_compact_rows: the requests now live in dense rows, use az_engine_advance_rows";
        return AZ_E_STATE;
    }
    rc = advance_range<false>(e, 0, e->p.G, priors, values, obs_out, stream, true);
    if (rc == AZ_OK) e->ticks++;
    return rc;
}

extern "C" int az_engine_compact_rows(az_engine *e, int32_t *n_live_out, void *stream) {
    if (!e || !n_live_out) return AZ_E_INVALID;
    if (!e->reset_done || e->ticks == 0) {
        e->err = "az_engine_compact_rows before the first tick";
        return AZ_E_STATE;
    }
    if (e->p.arena_agent != AZ_ARENA_SELF_PLAY || e->cfg.manual_moves) {
        e->err = "az_engine_compact_rows is for self-play engines (arena slots wait for each other; manual slots are re-armed)";
        return AZ_E_STATE;
    }
    unsigned long long next = 0;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipStreamSynchronize(st));
    HIPCHK(e, hipMemcpy(&next, e->p.next_game, sizeof next, hipMemcpyDeviceToHost));
    if ((long long)next < e->n_games) {
        e->err = "az_engine_compact_rows: games are still being handed out (idle slots would be refilled outside the list)";
        return AZ_E_STATE;
    }
    hipLaunchKernelGGL(az_compact_rows_kernel, dim3(1), dim3(1024), 0, st, e->p, e->rows_mapped ? 0 : 1);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(st));
    int n = 0;
    HIPCHK(e, hipMemcpy(&n, e->p.n_rows_live, sizeof n, hipMemcpyDeviceToHost));
    e->rows_mapped = true;
    e->rows_live = n;
    *n_live_out = n;
    return AZ_OK;
}

extern "C" int az_engine_advance_rows(az_engine *e, int32_t n_rows, const float *priors, const float *values, float *obs_out, void *stream) {
    int rc = advance_checks(e, obs_out);
    if (rc != AZ_OK) return rc;
    if (!e->rows_mapped || !priors || !values) {
        e->err = "az_engine_advance_rows needs az_engine_compact_rows first, and priors / values";
        return AZ_E_STATE;
    }
    if (n_rows < e->rows_live || n_rows > e->p.G) {
        e->err = "az_engine_advance_rows: n_rows must cover the live slots (az_engine_compact_rows' count) and fit n_slots";
        return AZ_E_INVALID;
    }
    if (e->rows_live == 0) return AZ_OK;
    rc = advance_range<true>(e, 0, e->rows_live, priors, values, obs_out, stream, true);
    if (rc == AZ_OK) e->ticks++;
    return rc;
}

extern "C" int az_engine_advance_slots(az_engine *e, int32_t first_slot, int32_t n_slots, const float *priors, const float *values,
                                       float *obs_out, void *stream) {
    int rc = advance_checks(e, obs_out);
    if (rc != AZ_OK) return rc;
    if (first_slot < 0 || n_slots < 1 || first_slot + n_slots > e->p.G || !priors || !values) {
        e->err = "az_engine_advance_slots: slot range outside [0, n_slots) or NULL priors/values";
        return AZ_E_INVALID;
    }
    if (e->rows_mapped) {
        e->err = "az_engine_advance_slots after az_engine_compact_rows: use az_engine_advance_rows";
        return AZ_E_STATE;
    }
    rc = advance_range<false>(e, first_slot, first_slot + n_slots, priors, values, obs_out, stream, false);
    if (rc == AZ_OK) e->ticks++;
    return rc;
}

extern "C" int az_engine_opponent_moves(az_engine *e, void *stream) {
    if (!e) return AZ_E_INVALID;
    if (e->p.arena_agent == AZ_ARENA_SELF_PLAY) {
        e->err = "az_engine_opponent_moves needs an engine created with arena_agent / arena_opponent";
        return AZ_E_STATE;
    }
    if (!e->reset_done) {
        e->err = "az_engine_opponent_moves before az_engine_reset";
        return AZ_E_STATE;
    }
    HIPCHK(e, hipSetDevice(e->cfg.device));
    dim3 grid((e->p.G + 63) / 64), block(64);
    if (e->cfg.game == AZ_GAME_CONNECT_FOUR) hipLaunchKernelGGL((az_opponent_kernel<AZG_CONNECT_FOUR>), grid, block, 0, (hipStream_t)stream, e->p);
    else hipLaunchKernelGGL((az_opponent_kernel<AZG_BREAKTHROUGH>), grid, block, 0, (hipStream_t)stream, e->p);
    HIPCHK(e, hipGetLastError());
    return AZ_OK;
}

// two engines facing each other: slot g of a and of b play the same game; a move the one has played (it is in its record
// store, and its root ply has moved past it) is handed to the other, which is waiting for exactly that ply
__global__ void az_exchange_kernel(Params a, Params b) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.G) return;
    if (a.phase[g] == PH_OPPONENT && b.ply[g] > a.ply[g]) {
        a.opp_action[g] = (int)b.rec_move[(size_t)g * b.max_plies + a.ply[g]];
        a.phase[g] = PH_OPP_DONE;
    }
    if (b.phase[g] == PH_OPPONENT && a.ply[g] > b.ply[g]) {
        b.opp_action[g] = (int)a.rec_move[(size_t)g * a.max_plies + b.ply[g]];
        b.phase[g] = PH_OPP_DONE;
    }
}

extern "C" int az_engine_exchange_moves(az_engine *a, az_engine *b, void *stream) {
    if (!a || !b) return AZ_E_INVALID;
    if (a->p.opp_kind != AZ_OPPONENT_EXTERNAL || b->p.opp_kind != AZ_OPPONENT_EXTERNAL || a->p.G != b->p.G || a->p.game != b->p.game ||
        a->p.geom.rows != b->p.geom.rows || a->p.geom.cols != b->p.geom.cols || a->p.arena_flip == b->p.arena_flip ||
        a->cfg.device != b->cfg.device || a->n_games != b->n_games || a->n_games > a->p.G) {
        a->err = "az_engine_exchange_moves: needs two AZ_OPPONENT_EXTERNAL engines of one game and device, equal n_slots, opposite "
                 "arena_flip, reset with the same n_games <= n_slots";
        return AZ_E_STATE;
    }
    if (!a->reset_done || !b->reset_done) {
        a->err = "az_engine_exchange_moves before az_engine_reset";
        return AZ_E_STATE;
    }
    HIPCHK(a, hipSetDevice(a->cfg.device));
    hipLaunchKernelGGL(az_exchange_kernel, dim3((a->p.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, a->p, b->p);
    HIPCHK(a, hipGetLastError());
    return AZ_OK;
}

extern "C" int az_engine_update_root(az_engine *e, const int32_t *actions, int32_t keep_subtree, void *stream) {
    if (!e || !actions) return AZ_E_INVALID;
    if (!e->cfg.manual_moves) {
        e->err = "az_engine_update_root needs an engine created with manual_moves = 1";
        return AZ_E_STATE;
    }
    if (!e->reset_done) {
        e->err = "az_engine_update_root before az_engine_reset";
        return AZ_E_STATE;
    }
    HIPCHK(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    if (!e->d_actions) HIPCHK(e, hipMalloc((void **)&e->d_actions, sizeof(int) * (size_t)e->p.G));
    for (int g = 0; g < e->p.G; g++)
        if (actions[g] >= e->p.A || actions[g] < AZ_ACTION_SEARCH_AGAIN) {
            e->err = "action out of range";
            return AZ_E_INVALID;
        }
    HIPCHK(e, hipStreamSynchronize(st)); // the staging buffer is reused call to call
    HIPCHK(e, hipMemcpy(e->d_actions, actions, sizeof(int) * (size_t)e->p.G, hipMemcpyHostToDevice));
    dim3 grid((e->p.G + 3) / 4), block(256);
    if (e->cfg.game == AZ_GAME_CONNECT_FOUR)
        hipLaunchKernelGGL((az_update_root_kernel<AZG_CONNECT_FOUR>), grid, block, 0, st, e->p, e->d_actions, keep_subtree);
    else
        hipLaunchKernelGGL((az_update_root_kernel<AZG_BREAKTHROUGH>), grid, block, 0, st, e->p, e->d_actions, keep_subtree);
    HIPCHK(e, hipGetLastError());
    return AZ_OK;
}

extern "C" int az_engine_progress(az_engine *e, az_progress *out, void *stream) {
    if (!e || !out) return AZ_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    size_t G = (size_t)e->p.G;
    std::vector<unsigned long long> stats(G * ST_N);
    std::vector<int> phase(G);
    unsigned long long next_game = 0, done = 0;
    unsigned int faults = 0;
    HIPCHK(e, hipStreamSynchronize(st));
    HIPCHK(e, hipMemcpy(stats.data(), e->p.stats, stats.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(phase.data(), e->p.phase, G * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&next_game, e->p.next_game, sizeof next_game, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&done, e->p.games_done, sizeof done, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&faults, e->p.faults, sizeof faults, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    for (size_t g = 0; g < G; g++) {
        const unsigned long long *s = &stats[g * ST_N];
        out->moves += (int64_t)s[ST_MOVES];
        out->sims += (int64_t)s[ST_SIMS];
        out->evals += (int64_t)s[ST_EVALS];
        out->terminal_hits += (int64_t)s[ST_TERM];
        out->sum_depth += (int64_t)s[ST_DEPTH];
        out->sum_children += (int64_t)s[ST_CHILDREN];
        out->nodes_allocated += (int64_t)s[ST_NODES];
        out->compactions += (int64_t)s[ST_COMPACT];
        if (phase[g] == PH_WAIT_LEAF || phase[g] == PH_WAIT_ROOT) out->slots_waiting++;
        if (phase[g] == PH_IDLE) out->slots_idle++;
        if (phase[g] == PH_SEARCH_DONE) out->slots_search_done++;
    }
    out->games_started = (int64_t)(next_game < (unsigned long long)e->n_games ? next_game : (unsigned long long)e->n_games);
    out->games_done = (int64_t)done;
    out->error_flags = faults;
    if (faults) {
        e->err = "device fault flags set:";
        if (faults & AZ_FAULT_POOL_EXHAUSTED) e->err += " POOL_EXHAUSTED";
        if (faults & AZ_FAULT_PLY_OVERFLOW) e->err += " PLY_OVERFLOW";
        if (faults & AZ_FAULT_NO_VISITS) e->err += " NO_VISITS";
        if (faults & AZ_FAULT_BAD_PRIOR) e->err += " BAD_PRIOR";
        if (faults & AZ_FAULT_ILLEGAL_ACTION) e->err += " ILLEGAL_ACTION";
        return AZ_E_DEVICE;
    }
    return AZ_OK;
}

extern "C" int az_engine_poll(az_engine *e, int64_t *games_done, uint32_t *error_flags, void *stream) {
    if (!e) return AZ_E_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long done = 0;
    unsigned int faults = 0;
    HIPCHK(e, hipMemcpyAsync(&done, e->p.games_done, sizeof done, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipMemcpyAsync(&faults, e->p.faults, sizeof faults, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
    if (games_done) *games_done = (int64_t)done;
    if (error_flags) *error_flags = faults;
    if (faults) {
        e->err = "device fault flags set (see az_engine_progress)";
        return AZ_E_DEVICE;
    }
    return AZ_OK;
}

extern "C" int az_engine_export(az_engine *e, az_example_view *out, void *stream) {
    if (!e || !out) return AZ_E_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipStreamSynchronize((hipStream_t)stream));
    size_t ng = (size_t)e->n_games, mp = (size_t)e->p.max_plies, mc = (size_t)e->p.maxc;
    e->h_len.resize(ng); e->h_ret0.resize(ng); e->h_states.resize(ng * mp * 2); e->h_move.resize(ng * mp);
    e->h_nchild.resize(ng * mp); e->h_child_action.resize(ng * mp * mc); e->h_child_visits.resize(ng * mp * mc);
    e->h_value.resize(ng * mp);
    HIPCHK(e, hipMemcpy(e->h_len.data(), e->p.rec_len, ng * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(e->h_ret0.data(), e->p.rec_ret0, ng * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(e->h_states.data(), e->p.rec_states, ng * mp * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(e->h_move.data(), e->p.rec_move, ng * mp * sizeof(uint16_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(e->h_nchild.data(), e->p.rec_nchild, ng * mp, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(e->h_child_action.data(), e->p.rec_child_action, ng * mp * mc * sizeof(uint16_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(e->h_child_visits.data(), e->p.rec_child_visits, ng * mp * mc * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(e->h_value.data(), e->p.rec_value, ng * mp * sizeof(double), hipMemcpyDeviceToHost));
    if (e->cfg.backup == AZ_BACKUP_ON_POLICY) // game_utils.py:200-204
        for (size_t g = 0; g < ng; g++) {
            double reward = (double)e->h_ret0[g];
            if (e->p.start.ply & 1) reward = -reward; // first recorded ply is player 1's
            for (int i = 0; i < e->h_len[g]; i++) {
                e->h_value[g * mp + (size_t)e->p.start.ply + i] = reward;
                reward *= -1;
            }
        }
    out->n_games = (int64_t)ng;
    out->max_plies = (int32_t)mp;
    out->max_children = (int32_t)mc;
    out->game_len = e->h_len.data();
    out->game_ret0 = e->h_ret0.data();
    out->states = e->h_states.data();
    out->move = e->h_move.data();
    out->n_children = e->h_nchild.data();
    out->child_action = e->h_child_action.data();
    out->child_visits = e->h_child_visits.data();
    out->value = e->h_value.data();
    return AZ_OK;
}

// ---- device-resident export: the generation's records packed into ONE caller-owned device buffer ----------------------
// layout (every array 16-byte aligned, n = games of the generation, mp = max_plies, mc = max_children):
//   game_len i32[n] | game_ret0 f32[n] | states u64[n][mp][2] | move u16[n][mp] | n_children u8[n][mp] |
//   child_action u16[n][mp][mc] | child_visits u32[n][mp][mc] | value f64[n][mp]
static void export_offsets(size_t n, size_t mp, size_t mc, size_t off[9]) {
    const size_t sizes[8] = {n * 4, n * 4, n * mp * 16, n * mp * 2, n * mp, n * mp * mc * 2, n * mp * mc * 4, n * mp * 8};
    off[0] = 0;
    for (int i = 0; i < 8; i++) off[i + 1] = off[i] + ((sizes[i] + 15) & ~(size_t)15);
}

__global__ void fill_on_policy_values_kernel(const int *len, const float *ret0, double *value, int n_games, int max_plies, int start_ply) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; // game_utils.py:200-204: z_i = returns()[0] * (-1)^i
    int g = (int)(t / max_plies), i = (int)(t % max_plies);
    if (g >= n_games || i >= len[g]) return;
    double z = (double)ret0[g];
    value[(size_t)g * max_plies + start_ply + i] = ((start_ply + i) & 1) ? -z : z;
}

extern "C" int64_t az_engine_export_device_bytes(const az_engine *e) {
    if (!e) return AZ_E_INVALID;
    size_t off[9];
    export_offsets((size_t)e->n_games, (size_t)e->p.max_plies, (size_t)e->p.maxc, off);
    return (int64_t)off[8];
}

extern "C" int az_engine_export_device(az_engine *e, void *dev_buf, int64_t bytes, void *stream) {
    if (!e || !dev_buf) return AZ_E_INVALID;
    size_t n = (size_t)e->n_games, mp = (size_t)e->p.max_plies, mc = (size_t)e->p.maxc, off[9];
    export_offsets(n, mp, mc, off);
    if (bytes < (int64_t)off[8]) {
        e->err = "az_engine_export_device: buffer smaller than az_engine_export_device_bytes()";
        return AZ_E_INVALID;
    }
    HIPCHK(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    if (e->cfg.backup == AZ_BACKUP_ON_POLICY) {
        long long threads = (long long)n * (long long)mp;
        hipLaunchKernelGGL(fill_on_policy_values_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, e->p.rec_len,
                           e->p.rec_ret0, e->p.rec_value, (int)n, (int)mp, e->p.start.ply);
        HIPCHK(e, hipGetLastError());
    }
    char *b = (char *)dev_buf;
    const void *src[8] = {e->p.rec_len, e->p.rec_ret0, e->p.rec_states, e->p.rec_move, e->p.rec_nchild, e->p.rec_child_action,
                          e->p.rec_child_visits, e->p.rec_value};
    const size_t sizes[8] = {n * 4, n * 4, n * mp * 16, n * mp * 2, n * mp, n * mp * mc * 2, n * mp * mc * 4, n * mp * 8};
    for (int i = 0; i < 8; i++) HIPCHK(e, hipMemcpyAsync(b + off[i], src[i], sizes[i], hipMemcpyDeviceToDevice, st));
    return AZ_OK;
}

extern "C" int az_engine_read_slot(az_engine *e, int32_t slot, az_slot_info *o) {
    if (!e || !o || slot < 0 || slot >= e->p.G) return AZ_E_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipDeviceSynchronize());
#define RD(dst, src) HIPCHK(e, hipMemcpy(&(dst), (src) + slot, sizeof(dst), hipMemcpyDeviceToHost))
    RD(o->phase, e->p.phase); RD(o->game_id, e->p.gid); RD(o->ply, e->p.ply); RD(o->sims_done, e->p.sims);
    RD(o->root, e->p.root); RD(o->alloc, e->p.alloc); RD(o->bb[0], e->p.bb0); RD(o->bb[1], e->p.bb1);
    RD(o->leaf_bb[0], e->p.leaf_bb0); RD(o->leaf_bb[1], e->p.leaf_bb1); RD(o->leaf_ply, e->p.leaf_ply);
    RD(o->depth, e->p.depth);
    o->depth &= 0xFFFF; // (the high half carries the action that leads to the requested leaf)
#undef RD
    return AZ_OK;
}

extern "C" int az_engine_read_root(az_engine *e, int32_t slot, int64_t *root_n, double *root_q, int32_t *actions,
                                   int64_t *child_n, double *child_q, double *child_p) {
    if (!e || slot < 0 || slot >= e->p.G) return AZ_E_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipDeviceSynchronize());
    int pool = 0;
    uint32_t root = 0, n = 0, c0 = 0, meta = 0;
    HIPCHK(e, hipMemcpy(&pool, e->p.which + slot, sizeof pool, hipMemcpyDeviceToHost));
    pool &= POOL_MASK; // bit 30 = the tree's select rule
    HIPCHK(e, hipMemcpy(&root, e->p.root + slot, sizeof root, hipMemcpyDeviceToHost));
    size_t base = (size_t)pool * e->p.cap;
    double q = 0;
    AzNode rn;
    HIPCHK(e, hipMemcpy(&rn, e->p.nodes + base + root, sizeof rn, hipMemcpyDeviceToHost));
    n = rn.N;
    q = rn.Q;
    c0 = rn.C0;
    meta = rn.META;
    if (root_n) *root_n = n;
    if (root_q) *root_q = q;
    int nc = c0 == NONE32 ? 0 : (int)(meta >> 16);
    if (nc > e->p.maxc) {
        e->err = "corrupt root";
        return AZ_E_DEVICE;
    }
    std::vector<AzNode> ch(nc);
    if (nc) HIPCHK(e, hipMemcpy(ch.data(), e->p.nodes + base + c0, sizeof(AzNode) * nc, hipMemcpyDeviceToHost));
    for (int i = 0; i < nc; i++) {
        if (actions) actions[i] = (int32_t)(ch[i].META & 0xFFFFu);
        if (child_n) child_n[i] = ch[i].N;
        if (child_q) child_q[i] = ch[i].Q;
        if (child_p) child_p[i] = ch[i].P;
    }
    return nc;
}

extern "C" int64_t az_engine_read_tree(az_engine *e, int32_t slot, int64_t max_nodes, int32_t *parent, int32_t *action,
                                       int64_t *n, double *q, double *pp) {
    if (!e || slot < 0 || slot >= e->p.G || max_nodes < 0) return AZ_E_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipDeviceSynchronize());
    int pool = 0;
    uint32_t root = 0, alloc = 0;
    HIPCHK(e, hipMemcpy(&pool, e->p.which + slot, sizeof pool, hipMemcpyDeviceToHost));
    pool &= POOL_MASK; // bit 30 = the tree's select rule
    HIPCHK(e, hipMemcpy(&root, e->p.root + slot, sizeof root, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&alloc, e->p.alloc + slot, sizeof alloc, hipMemcpyDeviceToHost));
    if (alloc > e->p.cap || root >= alloc) {
        e->err = "corrupt slot";
        return AZ_E_DEVICE;
    }
    size_t base = (size_t)pool * e->p.cap;
    std::vector<AzNode> nd_all(alloc);
    HIPCHK(e, hipMemcpy(nd_all.data(), e->p.nodes + base, sizeof(AzNode) * (size_t)alloc, hipMemcpyDeviceToHost));
    std::vector<uint32_t> N(alloc), C0(alloc), M(alloc);
    std::vector<double> Q(alloc), P(alloc);
    for (uint32_t i = 0; i < alloc; i++) {
        N[i] = nd_all[i].N;
        C0[i] = nd_all[i].C0;
        M[i] = nd_all[i].META;
        Q[i] = nd_all[i].Q;
        P[i] = nd_all[i].P;
    }
    std::vector<uint32_t> order;  // BFS queue of pool indices
    std::vector<int32_t> par;
    order.push_back(root);
    par.push_back(-1);
    for (size_t i = 0; i < order.size(); i++) {
        uint32_t nd = order[i];
        if (C0[nd] == NONE32) continue;
        uint32_t nc = M[nd] >> 16;
        if (C0[nd] + nc > alloc) {
            e->err = "corrupt tree";
            return AZ_E_DEVICE;
        }
        for (uint32_t k = 0; k < nc; k++) {
            order.push_back(C0[nd] + k);
            par.push_back((int32_t)i);
        }
    }
    int64_t cnt = (int64_t)order.size(), w = cnt < max_nodes ? cnt : max_nodes;
    for (int64_t i = 0; i < w; i++) {
        uint32_t nd = order[(size_t)i];
        if (parent) parent[i] = par[(size_t)i];
        if (action) action[i] = i == 0 ? -1 : (int32_t)(M[nd] & 0xFFFFu);
        if (n) n[i] = N[nd];
        if (q) q[i] = Q[nd];
        if (pp) pp[i] = P[nd];
    }
    return cnt;
}
