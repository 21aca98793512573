// Object-condensation loss reductions (reference: metrics/losses/oc.py:16-347,
// utils/graph_masks.py:19-28).
//
// Both reference variants - CondensationLossRG (radius graph) and CondensationLossTiger
// (dense N x K) - reduce to sums over (hit j, condensation point k) pairs:
//     attractive  sum_{j in particle k} q_j q_k |x_j - x_k|^2
//     repulsive   sum_{pid_j != pid_k, |x_j - x_k| < r} q_j q_k (r - sqrt(eps + |x_j - x_k|^2))
// plus two means over beta.  No N x K matrix and no radius graph is materialised: one
// thread owns a hit and streams the K condensation points through LDS (K ~ 1e3, a few KB);
// the gradient w.r.t. the condensation points is the transposed pass (thread = CP, hits
// streamed).  All sums are fixed-order (per-thread -> per-block fp64 partials -> one block):
// deterministic, no atomics.  Bound: fp32 VALU (N*K*D fma); HBM traffic is negligible.
//
// The radius-graph variant's `max_num_neighbors` cap is not emulated (torch_cluster keeps an
// implementation-defined subset when a hit has more neighbours than the cap; the reference
// tests never reach it) - see DESIGN.md.
#include "host_util.h"

namespace gnntrk {

typedef unsigned long long u64;
constexpr int kOcTpb = 256;
constexpr int kOcChunk = 128;  // condensation points staged per LDS pass
constexpr int kOcMaxDim = 32;

static int oc_grid(int64_t n) {
    int64_t g = ceil_div(n, kOcTpb);
    return (int)(g < 1 ? 1 : g);
}

__device__ __forceinline__ double oc_block_sum(double v, double *sh) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < kOcTpb / 64; ++i) s += sh[i];
    __syncthreads();
    return s;  // valid on thread 0
}

// utils/graph_masks.py:19-28
__global__ __launch_bounds__(kOcTpb) void good_node_mask_kernel(
    const float *__restrict__ pt, const int64_t *__restrict__ pid, const float *__restrict__ reco,
    const float *__restrict__ eta, int64_t n, float pt_thld, float max_eta,
    uint8_t *__restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    if (i < n)
        mask[i] = (pt[i] > pt_thld && pid[i] > 0 && reco[i] > 0.f && fabsf(eta[i]) < max_eta) ? 1 : 0;
}

// ---- condensation point selection (oc.py:16-43 / :279-292) ---------------------------
__global__ __launch_bounds__(kOcTpb) void oc_keys_kernel(const int64_t *__restrict__ pid, int64_t n,
                                                         u64 *__restrict__ keys,
                                                         uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    if (i < n) {
        keys[i] = (u64)pid[i];
        vals[i] = (uint32_t)i;
    }
}

// one thread per sorted position that starts a particle: best hit of the particle.
// mode 0 (RG):    candidates = masked hits, score = beta; qualifies iff a masked hit exists
// mode 1 (Tiger): candidates = all hits of the particle, score = q; qualifies iff any masked
// ties -> lowest hit index (the sort is stable, so hits of a particle are in index order)
__global__ __launch_bounds__(kOcTpb) void oc_segment_best_kernel(
    const u64 *__restrict__ keys, const uint32_t *__restrict__ order, const float *__restrict__ score,
    const uint8_t *__restrict__ mask, int64_t n, int mode, int32_t *__restrict__ seg_flag,
    int32_t *__restrict__ seg_best) {
    const int64_t s = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    if (s >= n) return;
    int32_t flag = 0, best = -1;
    if (s == 0 || keys[s] != keys[s - 1]) {
        const u64 key = keys[s];
        if (key != 0ull) {  // noise (pid 0) never is an object of interest (mask needs pid > 0)
            float bs = -1.f;
            bool any = false;
            for (int64_t t = s; t < n && keys[t] == key; ++t) {
                const uint32_t h = order[t];
                const bool m = mask[h] != 0;
                any = any || m;
                if ((mode == 1 || m) && score[h] > bs) {
                    bs = score[h];
                    best = (int32_t)h;
                }
            }
            flag = (any && best >= 0) ? 1 : 0;
        }
    }
    seg_flag[s] = flag;
    seg_best[s] = best;
}

// after the exclusive scan of seg_flag: alphas[k] = CP hit of the k-th particle of interest
// (ascending pid); gid[h] = k for every hit of that particle, else -1
__global__ __launch_bounds__(kOcTpb) void oc_assign_kernel(
    const u64 *__restrict__ keys, const uint32_t *__restrict__ order,
    const int32_t *__restrict__ seg_flag, const int32_t *__restrict__ seg_best,
    const int64_t *__restrict__ seg_off, int64_t n, int32_t *__restrict__ alphas,
    int32_t *__restrict__ gid, int32_t *__restrict__ n_cp) {
    const int64_t s = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    if (s >= n) return;
    if (s == 0) n_cp[0] = (int32_t)seg_off[n];
    if (s == 0 || keys[s] != keys[s - 1]) {
        const u64 key = keys[s];
        // gid is preset to -1: only particles of interest walk their (short) segment - the noise
        // "particle" (id 0) has a tenth of all hits and would keep one thread busy for milliseconds
        const int32_t k = seg_flag[s] ? (int32_t)seg_off[s] : -1;
        if (k >= 0) {
            alphas[k] = seg_best[s];
            for (int64_t t = s; t < n && keys[t] == key; ++t) gid[order[t]] = k;
        }
    }
}

// ---- potentials -----------------------------------------------------------------------
struct OcParams {
    const float *x;
    const float *beta;
    const int64_t *pid;
    const uint8_t *mask;
    const int32_t *gid;
    const int32_t *alphas;
    const int32_t *n_cp;
    int64_t n;
    int32_t dim, stride;
    float q_min, radius, eps_sqrt;
    int32_t mode;  // 0 RG, 1 Tiger
    float keep;    // probability of keeping a repulsive pair (>= 1: all)
    u64 seed;
    const int32_t *cap_nbr;  // RG neighbour cap: per hit its max_num_neighbors-th nearest hit inside the radius, -1: fewer
};

// The radius graph's neighbour cap (oc.py:115-117 radius_graph(max_num_neighbors)), nearest first: the
// condensation point a only repels hit j if a is among j's max_num_neighbors nearest hits, i.e.
// (d2(j, a), a) <= (d2(j, cap_j), cap_j) in the kNN search's own order and arithmetic (fmaf chain over
// the dimensions in order, ties to the lower index: csrc/knn.hip).
template <int DP>
__device__ __forceinline__ float oc_d2_chain(const float (&a)[DP], const float *b) {
    float d2 = 0.f;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        const float t = __fsub_rn(a[d], b[d]);
        d2 = __fmaf_rn(t, t, d2);
    }
    return d2;
}
__device__ __forceinline__ bool oc_cap_ok(float d2, int a, float cap_d2, int cap_idx) {
    return cap_idx < 0 || d2 < cap_d2 || (d2 == cap_d2 && a <= cap_idx);
}

// repulsive pair (hit j, condensation point k) kept?  splitmix64 of (seed, j, k) -> 24-bit uniform
__device__ __forceinline__ bool oc_keep_pair(const OcParams &p, int64_t j, int k) {
    if (p.keep >= 1.f) return true;
    u64 z = p.seed + 0x9E3779B97F4A7C15ull * ((u64)j * 0x100000001B3ull + (u64)(uint32_t)k + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.f / 16777216.f) < p.keep;
}

__device__ __forceinline__ float oc_q(float beta, float q_min) {
    const float a = atanhf(beta);
    return a * a + q_min;
}

// forward: per-block partial sums part[block][4] = {attractive, repulsive, n_rep_pairs, unused}
template <int DP>
__global__ __launch_bounds__(kOcTpb) void oc_forward_kernel(const OcParams p,
                                                            double *__restrict__ part) {
    __shared__ float s_x[kOcChunk][DP];
    __shared__ float s_q[kOcChunk];
    __shared__ long long s_pid[kOcChunk];
    __shared__ int s_hit[kOcChunk];  // hit index of the staged condensation point
    __shared__ double s_red[kOcTpb / 64];
    const int64_t j = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    const bool live = j < p.n;
    const int K = p.n_cp[0];
    float xj[DP];
    float qj = 0.f;
    long long pj = -1;
    int gj = -1;
    bool mj = false, is_cp = false;
#pragma unroll
    for (int d = 0; d < DP; ++d) xj[d] = (live && d < p.dim) ? p.x[j * p.stride + d] : 0.f;
    if (live) {
        qj = oc_q(p.beta[j], p.q_min);
        pj = p.pid[j];
        gj = p.gid[j];
        mj = p.mask[j] != 0;
        is_cp = gj >= 0 && p.alphas[gj] == (int32_t)j;
    }
    int cap_idx = -1;
    float cap_d2 = 0.f;
    if (live && p.cap_nbr) {
        cap_idx = p.cap_nbr[j];
        if (cap_idx >= 0) {
            float xc[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) xc[d] = d < p.dim ? p.x[(int64_t)cap_idx * p.stride + d] : 0.f;
            cap_d2 = oc_d2_chain<DP>(xj, xc);
        }
    }
    const float r2 = p.radius * p.radius;
    double va = 0.0, vr = 0.0, nrep = 0.0;
    for (int k0 = 0; k0 < K; k0 += kOcChunk) {
        __syncthreads();
        for (int i = threadIdx.x; i < kOcChunk; i += kOcTpb) {
            const int k = k0 + i;
            if (k < K) {
                const int32_t a = p.alphas[k];
#pragma unroll
                for (int d = 0; d < DP; ++d) s_x[i][d] = d < p.dim ? p.x[(int64_t)a * p.stride + d] : 0.f;
                s_q[i] = oc_q(p.beta[a], p.q_min);
                s_pid[i] = p.pid[a];
                s_hit[i] = a;
            }
        }
        __syncthreads();
        if (!live) continue;
        const int kn = (K - k0 < kOcChunk) ? (K - k0) : kOcChunk;
        for (int i = 0; i < kn; ++i) {
            float d2 = 0.f;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                const float t = xj[d] - s_x[i][d];
                d2 += t * t;
            }
            const float qq = qj * s_q[i];
            if (s_pid[i] == pj) {
                const bool att = (p.mode == 1) ? (gj == k0 + i) : (gj == k0 + i && mj && !is_cp);
                if (att) va += (double)(qq * d2);
            } else if (d2 < r2 && (cap_idx < 0 || oc_cap_ok(oc_d2_chain<DP>(xj, s_x[i]), s_hit[i], cap_d2, cap_idx))) {
                nrep += 1.0;  // (counted before the sub-sampling, as the reference's n_rep)
                if (oc_keep_pair(p, j, k0 + i)) vr += (double)(qq * (p.radius - sqrtf(p.eps_sqrt + d2)));
            }
        }
    }
    const double a = oc_block_sum(va, s_red), r = oc_block_sum(vr, s_red),
                 c = oc_block_sum(nrep, s_red);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 4 + 0] = a;
        part[blockIdx.x * 4 + 1] = r;
        part[blockIdx.x * 4 + 2] = c;
        part[blockIdx.x * 4 + 3] = 0.0;
    }
}

// out[0..3] = attractive, repulsive, coward, noise; out[4..7] = norm_att, norm_rep, K, n_rep
__global__ __launch_bounds__(kOcTpb) void oc_finalize_kernel(const OcParams p,
                                                             const double *__restrict__ part,
                                                             int n_part, float *__restrict__ out) {
    __shared__ double s_red[kOcTpb / 64];
    const int K = p.n_cp[0];
    double va = 0, vr = 0, nrep = 0, cow = 0, noise = 0, n_noise = 0, n_oi = 0;
    for (int i = threadIdx.x; i < n_part; i += kOcTpb) {
        va += part[i * 4 + 0];
        vr += part[i * 4 + 1];
        nrep += part[i * 4 + 2];
    }
    for (int k = threadIdx.x; k < K; k += kOcTpb) cow += (double)(1.f - p.beta[p.alphas[k]]);
    for (int64_t j = threadIdx.x; j < p.n; j += kOcTpb) {
        const bool is_noise = (p.mode == 1) ? !(p.pid[j] > 0) : (p.pid[j] == 0);
        if (is_noise) {
            noise += (double)p.beta[j];
            n_noise += 1.0;
        }
        if (p.mask[j]) n_oi += 1.0;
    }
    va = oc_block_sum(va, s_red);
    vr = oc_block_sum(vr, s_red);
    nrep = oc_block_sum(nrep, s_red);
    cow = oc_block_sum(cow, s_red);
    noise = oc_block_sum(noise, s_red);
    n_noise = oc_block_sum(n_noise, s_red);
    n_oi = oc_block_sum(n_oi, s_red);
    if (threadIdx.x == 0) {
        const double eps = 1e-9;
        const double norm_att = eps + n_oi - (double)K;
        double norm_rep = eps + ((double)K - 1.0) * (double)p.n;
        if (p.keep < 1.f) norm_rep *= (double)p.keep;  // oc.py:328
        out[0] = (float)(va / norm_att);
        out[1] = (float)(vr / norm_rep);
        out[2] = (float)(cow / (double)K);
        out[3] = (float)(noise / n_noise);
        out[4] = (float)norm_att;
        out[5] = (float)norm_rep;
        out[6] = (float)K;
        out[7] = (float)nrep;
        out[8] = (float)n_noise;
    }
}

// backward, hit side: gx[j] and gq[j] from the pairs of hit j; g = upstream grads of the 4
// (normalised) loss terms; norms from the forward's out[4..8]
template <int DP>
__global__ __launch_bounds__(kOcTpb) void oc_backward_hits_kernel(const OcParams p,
                                                                  const float *__restrict__ g,
                                                                  const float *__restrict__ fwd,
                                                                  float *__restrict__ gx,
                                                                  float *__restrict__ gbeta) {
    __shared__ float s_x[kOcChunk][DP];
    __shared__ float s_q[kOcChunk];
    __shared__ long long s_pid[kOcChunk];
    __shared__ int s_hit[kOcChunk];
    const int64_t j = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    const bool live = j < p.n;
    const int K = p.n_cp[0];
    const float ca = g[0] / fwd[4], cr = g[1] / fwd[5];
    float xj[DP], gxj[DP];
    float qj = 0.f, gqj = 0.f;
    long long pj = -1;
    int gj = -1;
    bool mj = false, is_cp = false;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        xj[d] = (live && d < p.dim) ? p.x[j * p.stride + d] : 0.f;
        gxj[d] = 0.f;
    }
    float bj = 0.5f;
    if (live) {
        bj = p.beta[j];
        qj = oc_q(bj, p.q_min);
        pj = p.pid[j];
        gj = p.gid[j];
        mj = p.mask[j] != 0;
        is_cp = gj >= 0 && p.alphas[gj] == (int32_t)j;
    }
    int cap_idx = -1;
    float cap_d2 = 0.f;
    if (live && p.cap_nbr) {
        cap_idx = p.cap_nbr[j];
        if (cap_idx >= 0) {
            float xc[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) xc[d] = d < p.dim ? p.x[(int64_t)cap_idx * p.stride + d] : 0.f;
            cap_d2 = oc_d2_chain<DP>(xj, xc);
        }
    }
    const float r2 = p.radius * p.radius;
    for (int k0 = 0; k0 < K; k0 += kOcChunk) {
        __syncthreads();
        for (int i = threadIdx.x; i < kOcChunk; i += kOcTpb) {
            const int k = k0 + i;
            if (k < K) {
                const int32_t a = p.alphas[k];
#pragma unroll
                for (int d = 0; d < DP; ++d) s_x[i][d] = d < p.dim ? p.x[(int64_t)a * p.stride + d] : 0.f;
                s_q[i] = oc_q(p.beta[a], p.q_min);
                s_pid[i] = p.pid[a];
                s_hit[i] = a;
            }
        }
        __syncthreads();
        if (!live) continue;
        const int kn = (K - k0 < kOcChunk) ? (K - k0) : kOcChunk;
        for (int i = 0; i < kn; ++i) {
            float t[DP];
            float d2 = 0.f;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                t[d] = xj[d] - s_x[i][d];
                d2 += t[d] * t[d];
            }
            float cx = 0.f, cq = 0.f;  // d/dx_j = cx * t ; d/dq_j = cq
            if (s_pid[i] == pj) {
                const bool att = (p.mode == 1) ? (gj == k0 + i) : (gj == k0 + i && mj && !is_cp);
                if (att) {
                    cx = ca * 2.f * qj * s_q[i];
                    cq = ca * s_q[i] * d2;
                }
            } else if (d2 < r2 && oc_keep_pair(p, j, k0 + i) &&
                       (cap_idx < 0 || oc_cap_ok(oc_d2_chain<DP>(xj, s_x[i]), s_hit[i], cap_d2, cap_idx))) {
                const float sd = sqrtf(p.eps_sqrt + d2);
                cx = sd > 0.f ? -cr * qj * s_q[i] / sd : 0.f;
                cq = cr * s_q[i] * (p.radius - sd);
            }
#pragma unroll
            for (int d = 0; d < DP; ++d) gxj[d] += cx * t[d];
            gqj += cq;
        }
    }
    if (live) {
        for (int d = 0; d < p.dim; ++d) gx[j * p.stride + d] = gxj[d];
        // q = atanh(beta)^2 + q_min ; plus the noise mean (the coward term is added by the CP pass)
        const float a = atanhf(bj);
        float gb = gqj * 2.f * a / (1.f - bj * bj);
        const bool is_noise = (p.mode == 1) ? !(pj > 0) : (pj == 0);
        if (is_noise) gb += g[3] / fwd[8];
        gbeta[j] = gb;
    }
}

// backward, condensation-point side: thread = CP k (of the batch [k_base, k_base + kOcCpBatch)),
// the hits are cut into gridDim.y slices streamed through LDS; every (slice, CP) writes its
// partial (d/dx_k [DP], d/dq_k) and oc_backward_cps_reduce_kernel adds the slices in slice order
// into gx[alpha_k], gbeta[alpha_k] (alpha_k are distinct hits: no conflicts).  Thread = CP alone
// would leave the chip idle: K is a few thousand (19 workgroups at K = 4625) while each of them
// walks all N hits.  Runs after the hit pass.
constexpr int kOcCpBatch = 16384;  // condensation points per launch (bounds the partial buffer)
constexpr int kOcSlices = 32;      // hit slices

template <int DP>
__global__ __launch_bounds__(kOcTpb) void oc_backward_cps_kernel(const OcParams p,
                                                                 const float *__restrict__ g,
                                                                 const float *__restrict__ fwd,
                                                                 int k_base, float *__restrict__ part) {
    __shared__ float s_x[kOcChunk][DP];
    __shared__ float s_q[kOcChunk];
    __shared__ long long s_pid[kOcChunk];
    __shared__ int s_att[kOcChunk];  // gid if the hit takes part in the attractive sum, else -2
    __shared__ int s_cap[kOcChunk];  // neighbour cap of the staged hit: index (-1: none) and distance
    __shared__ float s_capd2[kOcChunk];
    const int K = p.n_cp[0];
    if (k_base + (int)blockIdx.x * kOcTpb >= K) return;  // (uniform: before any barrier)
    const int kl = blockIdx.x * kOcTpb + threadIdx.x;    // CP inside the batch
    const int k = k_base + kl;
    const bool live = k < K;
    const float ca = g[0] / fwd[4], cr = g[1] / fwd[5];
    float xk[DP], gxk[DP];
    float qk = 0.f, gqk = 0.f;
    long long pk = -1;
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        xk[d] = 0.f;
        gxk[d] = 0.f;
    }
    int32_t ak = 0;
    if (live) {
        ak = p.alphas[k];
#pragma unroll
        for (int d = 0; d < DP; ++d) xk[d] = d < p.dim ? p.x[(int64_t)ak * p.stride + d] : 0.f;
        qk = oc_q(p.beta[ak], p.q_min);
        pk = p.pid[ak];
    }
    const float r2 = p.radius * p.radius;
    // slice of the hits (whole LDS chunks)
    const int64_t n_chunks = (p.n + kOcChunk - 1) / kOcChunk;
    const int64_t per = (n_chunks + gridDim.y - 1) / gridDim.y;
    const int64_t j_begin = (int64_t)blockIdx.y * per * kOcChunk;
    const int64_t j_stop = ((int64_t)(blockIdx.y + 1) * per * kOcChunk < p.n) ? (int64_t)(blockIdx.y + 1) * per * kOcChunk : p.n;
    for (int64_t j0 = j_begin; j0 < j_stop; j0 += kOcChunk) {
        __syncthreads();
        for (int i = threadIdx.x; i < kOcChunk; i += kOcTpb) {
            const int64_t j = j0 + i;
            if (j < p.n) {
#pragma unroll
                for (int d = 0; d < DP; ++d) s_x[i][d] = d < p.dim ? p.x[j * p.stride + d] : 0.f;
                s_q[i] = oc_q(p.beta[j], p.q_min);
                s_pid[i] = p.pid[j];
                const int gj = p.gid[j];
                bool att = gj >= 0;
                if (p.mode == 0) att = att && p.mask[j] != 0 && p.alphas[gj] != (int32_t)j;
                s_att[i] = att ? gj : -2;
                int ci = -1;
                float cd = 0.f;
                if (p.cap_nbr) {
                    ci = p.cap_nbr[j];
                    if (ci >= 0) {
                        float xa[DP];
#pragma unroll
                        for (int d = 0; d < DP; ++d) xa[d] = s_x[i][d];
                        float xc[DP];
#pragma unroll
                        for (int d = 0; d < DP; ++d) xc[d] = d < p.dim ? p.x[(int64_t)ci * p.stride + d] : 0.f;
                        cd = oc_d2_chain<DP>(xa, xc);
                    }
                }
                s_cap[i] = ci;
                s_capd2[i] = cd;
            }
        }
        __syncthreads();
        if (!live) continue;
        const int jn = (int)((p.n - j0 < kOcChunk) ? (p.n - j0) : kOcChunk);
        for (int i = 0; i < jn; ++i) {
            float t[DP];
            float d2 = 0.f;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                t[d] = s_x[i][d] - xk[d];  // x_j - x_k
                d2 += t[d] * t[d];
            }
            float cx = 0.f, cq = 0.f;  // d/dx_k = -cx * t ; d/dq_k = cq
            if (s_pid[i] == pk) {
                if (s_att[i] == k) {
                    cx = ca * 2.f * s_q[i] * qk;
                    cq = ca * s_q[i] * d2;
                }
            } else if (d2 < r2 && oc_keep_pair(p, j0 + i, k) &&
                       (s_cap[i] < 0 || oc_cap_ok(oc_d2_chain<DP>(s_x[i], xk), ak, s_capd2[i], s_cap[i]))) {
                const float sd = sqrtf(p.eps_sqrt + d2);
                cx = sd > 0.f ? -cr * s_q[i] * qk / sd : 0.f;
                cq = cr * s_q[i] * (p.radius - sd);
            }
#pragma unroll
            for (int d = 0; d < DP; ++d) gxk[d] -= cx * t[d];
            gqk += cq;
        }
    }
    if (live) {
        float *o = part + ((int64_t)blockIdx.y * kOcCpBatch + kl) * (DP + 1);
#pragma unroll
        for (int d = 0; d < DP; ++d) o[d] = gxk[d];
        o[DP] = gqk;
    }
}

template <int DP>
__global__ __launch_bounds__(kOcTpb) void oc_backward_cps_reduce_kernel(const OcParams p,
                                                                        const float *__restrict__ g,
                                                                        const float *__restrict__ fwd,
                                                                        int k_base, int n_slices,
                                                                        const float *__restrict__ part,
                                                                        float *__restrict__ gx,
                                                                        float *__restrict__ gbeta) {
    const int K = p.n_cp[0];
    const int kl = blockIdx.x * kOcTpb + threadIdx.x, k = k_base + kl;
    if (k >= K) return;
    float acc[DP + 1];
#pragma unroll
    for (int d = 0; d <= DP; ++d) acc[d] = 0.f;
    for (int sl = 0; sl < n_slices; ++sl) {  // fixed order: bit-reproducible
        const float *o = part + ((int64_t)sl * kOcCpBatch + kl) * (DP + 1);
#pragma unroll
        for (int d = 0; d <= DP; ++d) acc[d] += o[d];
    }
    const int32_t ak = p.alphas[k];
    for (int d = 0; d < p.dim; ++d) gx[(int64_t)ak * p.stride + d] += acc[d];
    const float bk = p.beta[ak];
    const float a = atanhf(bk);
    gbeta[ak] += acc[DP] * 2.f * a / (1.f - bk * bk) - g[2] / fwd[6];  // coward: mean(1 - beta)
}

// ---- the same sums without the N x K walk ("spatial" passes) ----------------------------------
// A repulsive pair needs |x_j - x_k| < radius, so almost all of the N x K pairs of a large event
// contribute exactly nothing.  The hits are sorted by Morton code into chunks of 64 with bounding
// boxes (knn.hip: spatial_chunks_build); a (chunk, condensation point) pair is only looked at if
// the box is closer to the point than the radius (lower bound sum_d max(lo_d - c_d, c_d - hi_d, 0)^2
// against radius^2 with a 1e-5 relative margin - far above the rounding of either side; the
// exact test d2 < radius^2 then runs unchanged).  The attractive term has no radius, but it is a
// sum over (hit, its own condensation point) only: it is taken outside the pair loops.
//   hit pass  (forward and backward): wave = chunk of 64 sorted hits (lane = hit), the K points
//             are tested 64 at a time (lane = point) and the survivors walked through LDS;
//   point pass (backward): wave = condensation point, the chunk boxes are tested 64 at a time
//             (lane = chunk), the surviving chunks streamed (lane = hit), its own particle's hits
//             come from a by-gid ordering; one fixed-order wave reduction, no partial buffers.
// Same pairs, same per-pair arithmetic as the dense kernels above; only the order of the
// (exactly representable) zero contributions that are skipped differs.
struct OcSpatial {
    const float *xs;       // [rows][DP] sorted hits
    const int32_t *sidx;   // [rows] original hit index, -1 in the tail
    const float *box;      // [n_chunks][2 DP]
    const float *hq;       // [rows] q of the sorted hit
    const long long *hpid; // [rows]
    const int32_t *hcap;   // [rows] neighbour cap of the sorted hit (-1: none) ...
    const float *hcapd2;   // ... and its distance
    const float *cx;       // [K][DP] condensation points
    const float *cq;       // [K]
    const long long *cpid; // [K]
    const uint32_t *gorder;   // hits ordered by gid (point pass)
    const uint32_t *gkeys;    // the sorted gid keys
    const int32_t *gstart;    // [K] first position of gid k in gorder
    int n_chunks;
};

template <int DP>
__global__ __launch_bounds__(kOcTpb) void oc_hit_records_kernel(const OcParams p, OcSpatial sp, float *__restrict__ hq,
                                                                long long *__restrict__ hpid,
                                                                int32_t *__restrict__ hcap,
                                                                float *__restrict__ hcapd2) {
    const int64_t r = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    if (r >= (int64_t)sp.n_chunks * 64) return;
    const int32_t j = sp.sidx[r];
    float q = 0.f, cd = 0.f;
    long long pid = -1;
    int32_t ci = -1;
    if (j >= 0) {
        q = oc_q(p.beta[j], p.q_min);
        pid = p.pid[j];
        if (p.cap_nbr) {
            ci = p.cap_nbr[j];
            if (ci >= 0) {
                float xa[DP], xc[DP];
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    xa[d] = sp.xs[r * DP + d];
                    xc[d] = d < p.dim ? p.x[(int64_t)ci * p.stride + d] : 0.f;
                }
                cd = oc_d2_chain<DP>(xa, xc);
            }
        }
    }
    hq[r] = q;
    hpid[r] = pid;
    hcap[r] = ci;
    hcapd2[r] = cd;
}

template <int DP>
__global__ __launch_bounds__(kOcTpb) void oc_cp_records_kernel(const OcParams p, float *__restrict__ cx,
                                                               float *__restrict__ cq, long long *__restrict__ cpid) {
    const int K = p.n_cp[0];
    const int k = blockIdx.x * kOcTpb + threadIdx.x;
    if (k >= K) return;
    const int32_t a = p.alphas[k];
#pragma unroll
    for (int d = 0; d < DP; ++d) cx[(int64_t)k * DP + d] = d < p.dim ? p.x[(int64_t)a * p.stride + d] : 0.f;
    cq[k] = oc_q(p.beta[a], p.q_min);
    cpid[k] = p.pid[a];
}

__device__ __forceinline__ void oc_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ double oc_wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float oc_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// hit pass: workgroup = chunk of 64 sorted hits, its four waves take every fourth round of 64
// condensation points (the rounds are latency bound: more of them in flight) and their per-hit
// sums are added in wave order through LDS.  BWD = false: part[chunk][8] = {attractive, repulsive,
// n_rep_pairs, sum of beta over noise hits, noise hits, hits of interest, 0, 0}
// (oc_finalize_spatial_kernel adds the chunks); BWD = true: gx[j], gbeta[j] of every hit (the point
// pass adds the condensation points' share)
constexpr int kOcHitWaves = kOcTpb / 64;
template <int DP, bool BWD>
__global__ __launch_bounds__(kOcTpb) void oc_hits_spatial_kernel(const OcParams p, const OcSpatial sp,
                                                                 const float *__restrict__ g,
                                                                 const float *__restrict__ fwd,
                                                                 double *__restrict__ part, float *__restrict__ gx,
                                                                 float *__restrict__ gbeta) {
    __shared__ float s_x[kOcHitWaves][64][DP];
    __shared__ float s_q[kOcHitWaves][64];
    __shared__ long long s_pid[kOcHitWaves][64];
    __shared__ int s_hit[kOcHitWaves][64];
    __shared__ double s_accd[kOcHitWaves][2][64];     // forward: repulsive, pair count
    __shared__ float s_accf[kOcHitWaves][DP + 1][64];  // backward: d/dx, d/dq
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const int K = p.n_cp[0];
    const int64_t r = (int64_t)c * 64 + lane;
    const int32_t jj = sp.sidx[r];
    const bool live = jj >= 0;
    const int64_t j = live ? jj : 0;
    float xj[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) xj[d] = sp.xs[r * DP + d];
    const float qj = sp.hq[r];
    const long long pj = sp.hpid[r];
    const int cap_idx = sp.hcap[r];
    const float cap_d2 = sp.hcapd2[r];
    float blo[DP], bhi[DP];  // the chunk's box (wave-uniform)
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        blo[d] = sp.box[(int64_t)c * 2 * DP + d];
        bhi[d] = sp.box[(int64_t)c * 2 * DP + DP + d];
    }
    const float r2 = p.radius * p.radius;
    const float r2m = r2 * 1.00001f + 1e-30f;
    float ca = 0.f, cr = 0.f;
    if (BWD) {
        ca = g[0] / fwd[4];
        cr = g[1] / fwd[5];
    }
    double vr = 0.0, nrep = 0.0;
    float gxj[DP];
    float gqj = 0.f;
#pragma unroll
    for (int d = 0; d < DP; ++d) gxj[d] = 0.f;

    // lane = condensation point: distance bound to the chunk's box; the next round's coordinates
    // are in flight while the survivors of this one are walked
    auto load_cp = [&](int k0, float (&ck)[DP]) {
        const int k = k0 + lane;
        const int kc = k < K ? k : K - 1;
#pragma unroll
        for (int d = 0; d < DP; ++d) ck[d] = sp.cx[(int64_t)kc * DP + d];
    };
    float ck[DP], cn[DP];
    load_cp(wv * 64, ck);
    for (int k0 = wv * 64; k0 < K; k0 += kOcHitWaves * 64) {
        load_cp(k0 + kOcHitWaves * 64 < K ? k0 + kOcHitWaves * 64 : k0, cn);
        const int k = k0 + lane;
        const int kc = k < K ? k : K - 1;
        float lb = 0.f;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const float gd = fmaxf(fmaxf(blo[d] - ck[d], ck[d] - bhi[d]), 0.f);
            lb += gd * gd;
        }
        unsigned long long near = __ballot(k < K && lb <= r2m);
        if (near != 0ull) {
            oc_wave_sync();  // (the previous round's readers are done)
#pragma unroll
            for (int d = 0; d < DP; ++d) s_x[wv][lane][d] = ck[d];
            s_q[wv][lane] = sp.cq[kc];
            s_pid[wv][lane] = sp.cpid[kc];
            s_hit[wv][lane] = p.alphas[kc];
            oc_wave_sync();
            while (near != 0ull) {
                const int i = __ffsll(near) - 1;
                near &= near - 1ull;
                float t[DP];
                float d2 = 0.f;
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    t[d] = xj[d] - s_x[wv][i][d];
                    d2 += t[d] * t[d];
                }
                if (live && s_pid[wv][i] != pj && d2 < r2 &&
                    (cap_idx < 0 || oc_cap_ok(oc_d2_chain<DP>(xj, s_x[wv][i]), s_hit[wv][i], cap_d2, cap_idx))) {
                    const float qk = s_q[wv][i];
                    if (!BWD) {
                        nrep += 1.0;  // (counted before the sub-sampling, as the reference's n_rep)
                        if (oc_keep_pair(p, j, k0 + i)) vr += (double)(qj * qk * (p.radius - sqrtf(p.eps_sqrt + d2)));
                    } else if (oc_keep_pair(p, j, k0 + i)) {
                        const float sd = sqrtf(p.eps_sqrt + d2);
                        const float cxr = sd > 0.f ? -cr * qj * qk / sd : 0.f;
#pragma unroll
                        for (int d = 0; d < DP; ++d) gxj[d] += cxr * t[d];
                        gqj += cr * qk * (p.radius - sd);
                    }
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DP; ++d) ck[d] = cn[d];
    }
    if (!BWD) {
        s_accd[wv][0][lane] = vr;
        s_accd[wv][1][lane] = nrep;
    } else {
#pragma unroll
        for (int d = 0; d < DP; ++d) s_accf[wv][d][lane] = gxj[d];
        s_accf[wv][DP][lane] = gqj;
    }
    __syncthreads();
    if (wv != 0) return;
    if (!BWD) {
        vr = ((s_accd[0][0][lane] + s_accd[1][0][lane]) + s_accd[2][0][lane]) + s_accd[3][0][lane];
        nrep = ((s_accd[0][1][lane] + s_accd[1][1][lane]) + s_accd[2][1][lane]) + s_accd[3][1][lane];
    } else {
#pragma unroll
        for (int d = 0; d < DP; ++d)
            gxj[d] = ((s_accf[0][d][lane] + s_accf[1][d][lane]) + s_accf[2][d][lane]) + s_accf[3][d][lane];
        gqj = ((s_accf[0][DP][lane] + s_accf[1][DP][lane]) + s_accf[2][DP][lane]) + s_accf[3][DP][lane];
    }
    int gj = -1;
    bool mj = false, is_cp = false;
    float bj = 0.5f;
    if (live) {
        gj = p.gid[j];
        mj = p.mask[j] != 0;
        is_cp = gj >= 0 && p.alphas[gj] == jj;
        bj = p.beta[j];
    }
    // attractive: the hit and the condensation point of its own particle, at any distance
    double va = 0.0;
    const bool att = live && gj >= 0 && ((p.mode == 1) ? true : (mj && !is_cp));
    if (att) {
        float t[DP];
        float d2 = 0.f;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            t[d] = xj[d] - sp.cx[(int64_t)gj * DP + d];
            d2 += t[d] * t[d];
        }
        const float qk = sp.cq[gj];
        if (!BWD) {
            va += (double)(qj * qk * d2);
        } else {
            const float cxa = ca * 2.f * qj * qk;
#pragma unroll
            for (int d = 0; d < DP; ++d) gxj[d] += cxa * t[d];
            gqj += ca * qk * d2;
        }
    }
    const bool is_noise = live && ((p.mode == 1) ? !(pj > 0) : (pj == 0));
    if (!BWD) {
        const double a = oc_wave_sum(va), rr = oc_wave_sum(vr), cc = oc_wave_sum(nrep);
        const double ns = oc_wave_sum(is_noise ? (double)bj : 0.0), nn = oc_wave_sum(is_noise ? 1.0 : 0.0),
                     no = oc_wave_sum(mj ? 1.0 : 0.0);
        if (lane == 0) {
            double *o = part + (int64_t)c * 8;
            o[0] = a; o[1] = rr; o[2] = cc; o[3] = ns; o[4] = nn; o[5] = no; o[6] = 0.0; o[7] = 0.0;
        }
    } else if (live) {
        for (int d = 0; d < p.dim; ++d) gx[j * p.stride + d] = gxj[d];
        const float a = atanhf(bj);
        float gb = gqj * 2.f * a / (1.f - bj * bj);
        if (is_noise) gb += g[3] / fwd[8];
        gbeta[j] = gb;
    }
}

// the chunks' partial sums -> out[0..8] (as oc_finalize_kernel, without its walk over all hits)
__global__ __launch_bounds__(kOcTpb) void oc_finalize_spatial_kernel(const OcParams p, const double *__restrict__ part,
                                                                     int n_part, float *__restrict__ out) {
    __shared__ double s_red[kOcTpb / 64];
    const int K = p.n_cp[0];
    double v[6] = {0, 0, 0, 0, 0, 0}, cow = 0;
    for (int i = threadIdx.x; i < n_part; i += kOcTpb)
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] += part[(int64_t)i * 8 + u];
    for (int k = threadIdx.x; k < K; k += kOcTpb) cow += (double)(1.f - p.beta[p.alphas[k]]);
#pragma unroll
    for (int u = 0; u < 6; ++u) v[u] = oc_block_sum(v[u], s_red);
    cow = oc_block_sum(cow, s_red);
    if (threadIdx.x == 0) {
        const double eps = 1e-9;
        const double norm_att = eps + v[5] - (double)K;
        double norm_rep = eps + ((double)K - 1.0) * (double)p.n;
        if (p.keep < 1.f) norm_rep *= (double)p.keep;  // oc.py:328
        out[0] = (float)(v[0] / norm_att);
        out[1] = (float)(v[1] / norm_rep);
        out[2] = (float)(cow / (double)K);
        out[3] = (float)(v[3] / v[4]);
        out[4] = (float)norm_att;
        out[5] = (float)norm_rep;
        out[6] = (float)K;
        out[7] = (float)v[2];
        out[8] = (float)v[4];
    }
}

// gstart[k] = first position of gid k in the by-gid ordering (every k < K owns at least its
// condensation point)
__global__ __launch_bounds__(kOcTpb) void oc_gid_keys_kernel(const int32_t *__restrict__ gid, int64_t n,
                                                             uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    if (i < n) {
        keys[i] = (uint32_t)gid[i];  // -1 -> 0xffffffff: hits of no particle of interest sort last
        vals[i] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(kOcTpb) void oc_gid_starts_kernel(const uint32_t *__restrict__ keys, int64_t n,
                                                               int32_t *__restrict__ gstart) {
    const int64_t s = (int64_t)blockIdx.x * kOcTpb + threadIdx.x;
    if (s >= n) return;
    const uint32_t key = keys[s];
    if (key != 0xffffffffu && (s == 0 || keys[s - 1] != key)) gstart[key] = (int32_t)s;
}

// point pass (backward): workgroup = condensation point k; its four waves take every fourth batch of
// 64 chunk boxes, wave 0 also the particle's own hits; fixed-order reductions (lanes, then waves)
template <int DP>
__global__ __launch_bounds__(kOcTpb) void oc_cps_spatial_kernel(const OcParams p, const OcSpatial sp,
                                                                const float *__restrict__ g,
                                                                const float *__restrict__ fwd,
                                                                float *__restrict__ gx, float *__restrict__ gbeta) {
    __shared__ float s_acc[kOcHitWaves][DP + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int K = p.n_cp[0];
    const int k = blockIdx.x;
    if (k >= K) return;  // (workgroup-uniform)
    const float ca = g[0] / fwd[4], cr = g[1] / fwd[5];
    const int32_t ak = p.alphas[k];
    float xk[DP], gxk[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        xk[d] = sp.cx[(int64_t)k * DP + d];
        gxk[d] = 0.f;
    }
    const float qk = sp.cq[k];
    const long long pk = sp.cpid[k];
    float gqk = 0.f;
    const float r2 = p.radius * p.radius;
    const float r2m = r2 * 1.00001f + 1e-30f;
    // repulsive: the chunks whose box reaches into the point's radius
    for (int c0 = wv * 64; c0 < sp.n_chunks; c0 += kOcHitWaves * 64) {
        const int c = c0 + lane;
        const int cc = c < sp.n_chunks ? c : sp.n_chunks - 1;
        float lb = 0.f;
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            const float lo = sp.box[(int64_t)cc * 2 * DP + d], hi = sp.box[(int64_t)cc * 2 * DP + DP + d];
            const float gd = fmaxf(fmaxf(lo - xk[d], xk[d] - hi), 0.f);
            lb += gd * gd;
        }
        unsigned long long near = __ballot(c < sp.n_chunks && lb <= r2m);
        if (near == 0ull) continue;
        // the next surviving chunk's rows are in flight while this one is evaluated
        float xa[DP], xn[DP];
        int32_t jj, jn;
        long long pj, pn;
        float qj, qn;
        auto load_rows = [&](int i, float (&x)[DP], int32_t &id, long long &pid, float &q) {
            const int64_t r = (int64_t)(c0 + i) * 64 + lane;
#pragma unroll
            for (int d = 0; d < DP; ++d) x[d] = sp.xs[r * DP + d];
            id = sp.sidx[r];
            pid = sp.hpid[r];
            q = sp.hq[r];
        };
        int i = __ffsll(near) - 1;
        load_rows(i, xa, jj, pj, qj);
        while (near != 0ull) {
            near &= near - 1ull;
            const int in = near != 0ull ? __ffsll(near) - 1 : i;
            load_rows(in, xn, jn, pn, qn);
            const int64_t r = (int64_t)(c0 + i) * 64 + lane;
            float t[DP];
            float d2 = 0.f;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                t[d] = xa[d] - xk[d];  // x_j - x_k
                d2 += t[d] * t[d];
            }
            if (jj >= 0 && pj != pk && d2 < r2 && oc_keep_pair(p, jj, k)) {
                const int ci = sp.hcap[r];
                if (ci < 0 || oc_cap_ok(oc_d2_chain<DP>(xa, xk), ak, sp.hcapd2[r], ci)) {
                    const float sd = sqrtf(p.eps_sqrt + d2);
                    const float cxr = sd > 0.f ? -cr * qj * qk / sd : 0.f;
#pragma unroll
                    for (int d = 0; d < DP; ++d) gxk[d] -= cxr * t[d];
                    gqk += cr * qj * (p.radius - sd);
                }
            }
#pragma unroll
            for (int d = 0; d < DP; ++d) xa[d] = xn[d];
            jj = jn;
            pj = pn;
            qj = qn;
            i = in;
        }
    }
    // attractive: the hits of particle k (by-gid ordering), at any distance
    if (wv == 0) {
        for (int64_t s = sp.gstart[k];; s += 64) {
            const int64_t t0 = s + lane;
            const bool mine = t0 < p.n && sp.gkeys[t0 < p.n ? t0 : p.n - 1] == (uint32_t)k;
            if (mine) {
                const int64_t j = sp.gorder[t0];
                const bool att = (p.mode == 1) ? true : (p.mask[j] != 0 && ak != (int32_t)j);
                if (att) {
                    float t[DP];
                    float d2 = 0.f;
#pragma unroll
                    for (int d = 0; d < DP; ++d) {
                        t[d] = (d < p.dim ? p.x[j * p.stride + d] : 0.f) - xk[d];
                        d2 += t[d] * t[d];
                    }
                    const float qj = oc_q(p.beta[j], p.q_min);
                    const float cxa = ca * 2.f * qj * qk;
#pragma unroll
                    for (int d = 0; d < DP; ++d) gxk[d] -= cxa * t[d];
                    gqk += ca * qj * d2;
                }
            }
            if (__ballot(mine) != ~0ull) break;  // the run of gid k ended inside this step
        }
    }
#pragma unroll
    for (int d = 0; d < DP; ++d) gxk[d] = oc_wave_sum(gxk[d]);
    gqk = oc_wave_sum(gqk);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < DP; ++d) s_acc[wv][d] = gxk[d];
        s_acc[wv][DP] = gqk;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int d = 0; d < p.dim; ++d)
            gx[(int64_t)ak * p.stride + d] += ((s_acc[0][d] + s_acc[1][d]) + s_acc[2][d]) + s_acc[3][d];
        const float gq = ((s_acc[0][DP] + s_acc[1][DP]) + s_acc[2][DP]) + s_acc[3][DP];
        const float bk = p.beta[ak];
        const float a = atanhf(bk);
        gbeta[ak] += gq * 2.f * a / (1.f - bk * bk) - g[2] / fwd[6];  // coward: mean(1 - beta)
    }
}

// ---- launchers -----------------------------------------------------------------------
size_t sort_pairs_u64_temp_bytes(int64_t n);
int sort_pairs_u64(const u64 *keys_in, u64 *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                   int64_t n, void *temp, size_t temp_bytes, hipStream_t stream);

int good_node_mask_launch(const float *pt, const int64_t *pid, const float *reco, const float *eta,
                          int64_t n, float pt_thld, float max_eta, uint8_t *mask,
                          hipStream_t stream) {
    if (n < 0) return fail(GNNTRK_EINVAL, "good_node_mask: bad argument");
    if (n == 0) return GNNTRK_OK;
    if (!pt || !pid || !reco || !eta || !mask) return fail(GNNTRK_EINVAL, "good_node_mask: NULL pointer");
    hipLaunchKernelGGL(good_node_mask_kernel, dim3(oc_grid(n)), dim3(kOcTpb), 0, stream, pt, pid, reco,
                       eta, n, pt_thld, max_eta, mask);
    return check_launch("good_node_mask");
}

size_t oc_select_ws_bytes(int64_t n) {
    const size_t a8 = align_up((size_t)(n > 0 ? n : 1) * 8, 256);
    const size_t a4 = align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    return 2 * a8 /*keys*/ + 2 * a4 /*vals*/ + 2 * a4 /*flag,best*/ + align_up((size_t)(n + 1) * 8, 256) +
           align_up(sort_pairs_u64_temp_bytes(n), 256);
}

int oc_select_launch(const float *score, const int64_t *pid, const uint8_t *mask, int64_t n, int mode,
                     int32_t *alphas, int32_t *gid, int32_t *n_cp, void *ws, size_t ws_bytes,
                     hipStream_t stream) {
    if (!score || !pid || !mask || !alphas || !gid || !n_cp || n < 1 || (mode != 0 && mode != 1))
        return fail(GNNTRK_EINVAL, "oc_select_cps: bad argument");
    if (n > 0x7fffffff) return fail(GNNTRK_EUNSUPPORTED, "oc_select_cps: n must fit int32");
    if (!ws || ws_bytes < oc_select_ws_bytes(n)) return fail(GNNTRK_EINVAL, "oc_select_cps: workspace too small");
    char *w = reinterpret_cast<char *>(ws);
    const size_t a8 = align_up((size_t)n * 8, 256), a4 = align_up((size_t)n * 4, 256);
    u64 *keys_a = (u64 *)w; w += a8;
    u64 *keys_b = (u64 *)w; w += a8;
    uint32_t *vals_a = (uint32_t *)w; w += a4;
    uint32_t *vals_b = (uint32_t *)w; w += a4;
    int32_t *flag = (int32_t *)w; w += a4;
    int32_t *best = (int32_t *)w; w += a4;
    int64_t *off = (int64_t *)w; w += align_up((size_t)(n + 1) * 8, 256);
    void *temp = w;
    const int grid = oc_grid(n);
    hipLaunchKernelGGL(oc_keys_kernel, dim3(grid), dim3(kOcTpb), 0, stream, pid, n, keys_a, vals_a);
    int rc = sort_pairs_u64(keys_a, keys_b, vals_a, vals_b, n, temp, sort_pairs_u64_temp_bytes(n), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(oc_segment_best_kernel, dim3(grid), dim3(kOcTpb), 0, stream,
                       (const u64 *)keys_b, (const uint32_t *)vals_b, score, mask, n, mode, flag, best);
    scan_counts_launch(flag, 0x7fffffff, n, off, stream);
    rc = check_hip(hipMemsetAsync(gid, 0xff, (size_t)n * sizeof(int32_t), stream), "oc_select_cps(memset)");
    if (rc) return rc;
    hipLaunchKernelGGL(oc_assign_kernel, dim3(grid), dim3(kOcTpb), 0, stream, (const u64 *)keys_b,
                       (const uint32_t *)vals_b, (const int32_t *)flag, (const int32_t *)best,
                       (const int64_t *)off, n, alphas, gid, n_cp);
    return check_launch("oc_select_cps");
}

static int oc_check(const gnntrk_oc_args *a) {
    if (!a || !a->x || !a->beta || !a->particle_id || !a->mask || !a->gid || !a->alphas || !a->n_cp)
        return fail(GNNTRK_EINVAL, "oc_potential: NULL pointer");
    if (a->n < 1 || a->dim < 1 || a->dim > kOcMaxDim || a->stride < a->dim)
        return fail(GNNTRK_EINVAL, "oc_potential: bad sizes (1 <= dim <= 32)");
    if (a->mode != 0 && a->mode != 1) return fail(GNNTRK_EINVAL, "oc_potential: mode must be 0 or 1");
    return GNNTRK_OK;
}

static OcParams oc_params(const gnntrk_oc_args *a) {
    OcParams p;
    p.x = a->x; p.beta = a->beta; p.pid = a->particle_id; p.mask = a->mask; p.gid = a->gid;
    p.alphas = a->alphas; p.n_cp = a->n_cp; p.n = a->n; p.dim = a->dim; p.stride = a->stride;
    p.q_min = a->q_min; p.radius = a->radius; p.eps_sqrt = a->eps_sqrt; p.mode = a->mode;
    p.keep = a->rep_keep_prob > 0.f ? a->rep_keep_prob : 1.f;   // (0 = field left unset: all pairs)
    p.seed = a->rep_seed;
    p.cap_nbr = a->cap_nbr;
    return p;
}

size_t oc_forward_ws_bytes(int64_t n) { return (size_t)oc_grid(n) * 4 * sizeof(double); }

#define OC_DISPATCH(CALL)                \
    if (a->dim <= 2) { CALL(2); }        \
    else if (a->dim <= 4) { CALL(4); }   \
    else if (a->dim <= 8) { CALL(8); }   \
    else if (a->dim <= 16) { CALL(16); } \
    else { CALL(32); }

int oc_forward_launch(const gnntrk_oc_args *a, float *out, void *ws, size_t ws_bytes, hipStream_t stream) {
    int rc = oc_check(a);
    if (rc) return rc;
    if (!out || !ws || ws_bytes < oc_forward_ws_bytes(a->n)) return fail(GNNTRK_EINVAL, "oc_forward: workspace too small");
    const OcParams p = oc_params(a);
    const int grid = oc_grid(a->n);
    double *part = reinterpret_cast<double *>(ws);
#define CALL_F(DP) hipLaunchKernelGGL(oc_forward_kernel<DP>, dim3(grid), dim3(kOcTpb), 0, stream, p, part)
    OC_DISPATCH(CALL_F)
#undef CALL_F
    hipLaunchKernelGGL(oc_finalize_kernel, dim3(1), dim3(kOcTpb), 0, stream, p, (const double *)part, grid, out);
    return check_launch("oc_forward");
}

// partial buffer of the condensation-point pass: [slices][kOcCpBatch][dim padded + 1] floats
size_t oc_backward_ws_bytes(int64_t n, int dim) {
    (void)n;
    const int dp = dim <= 2 ? 2 : dim <= 4 ? 4 : dim <= 8 ? 8 : dim <= 16 ? 16 : 32;
    return (size_t)kOcSlices * kOcCpBatch * (dp + 1) * sizeof(float);
}

int oc_backward_launch(const gnntrk_oc_args *a, const float *g, const float *fwd, float *gx, float *gbeta,
                       int64_t max_cps, void *ws, size_t ws_bytes, hipStream_t stream) {
    int rc = oc_check(a);
    if (rc) return rc;
    if (!g || !fwd || !gx || !gbeta || max_cps < 1) return fail(GNNTRK_EINVAL, "oc_backward: bad argument");
    if (!ws || ws_bytes < oc_backward_ws_bytes(a->n, a->dim)) return fail(GNNTRK_EINVAL, "oc_backward: workspace too small");
    const OcParams p = oc_params(a);
    const int grid = oc_grid(a->n);
#define CALL_BH(DP) hipLaunchKernelGGL(oc_backward_hits_kernel<DP>, dim3(grid), dim3(kOcTpb), 0, stream, p, g, fwd, gx, gbeta)
    OC_DISPATCH(CALL_BH)
#undef CALL_BH
    // condensation-point side in batches of kOcCpBatch (the count K lives on the device: batches
    // past it exit at once), each batch cut into hit slices
    int64_t n_chunks = (a->n + kOcChunk - 1) / kOcChunk;
    const int slices = (int)(n_chunks < kOcSlices ? n_chunks : kOcSlices);
    float *part = reinterpret_cast<float *>(ws);
    for (int64_t kb = 0; kb < max_cps; kb += kOcCpBatch) {
        const int64_t in_batch = (max_cps - kb < kOcCpBatch) ? (max_cps - kb) : kOcCpBatch;
        const int kgrid = oc_grid(in_batch);
#define CALL_BC(DP)                                                                                          \
    hipLaunchKernelGGL(oc_backward_cps_kernel<DP>, dim3(kgrid, slices), dim3(kOcTpb), 0, stream, p, g, fwd,    \
                       (int)kb, part);                                                                        \
    hipLaunchKernelGGL(oc_backward_cps_reduce_kernel<DP>, dim3(kgrid), dim3(kOcTpb), 0, stream, p, g, fwd,     \
                       (int)kb, slices, (const float *)part, gx, gbeta)
        OC_DISPATCH(CALL_BC)
#undef CALL_BC
    }
    return check_launch("oc_backward");
}

// ---- spatial passes: workspace layout and launchers ------------------------------------------
struct OcSpatialWs {
    size_t xs, sidx, box, hq, hpid, hcap, hcapd2, cx, cq, cpid, part, gkeys_a, gkeys_b, gvals_a, gvals_b, gstart,
        gtemp, scratch, total;
    int n_chunks, dp;
};
static OcSpatialWs oc_spatial_layout(int64_t n, int dim) {
    OcSpatialWs w{};
    w.dp = spatial_dp(dim);
    w.n_chunks = spatial_n_chunks(n);
    const size_t rows = (size_t)w.n_chunks * 64;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += align_up(bytes, 256);
        return at;
    };
    w.xs = take(rows * w.dp * 4);
    w.sidx = take(rows * 4);
    w.box = take((size_t)w.n_chunks * 2 * w.dp * 4);
    w.hq = take(rows * 4);
    w.hpid = take(rows * 8);
    w.hcap = take(rows * 4);
    w.hcapd2 = take(rows * 4);
    w.cx = take((size_t)n * w.dp * 4);
    w.cq = take((size_t)n * 4);
    w.cpid = take((size_t)n * 8);
    w.part = take((size_t)w.n_chunks * 8 * sizeof(double));
    w.gkeys_a = take((size_t)n * 4);
    w.gkeys_b = take((size_t)n * 4);
    w.gvals_a = take((size_t)n * 4);
    w.gvals_b = take((size_t)n * 4);
    w.gstart = take((size_t)n * 4);
    w.gtemp = take(sort_pairs_temp_bytes(n));
    w.scratch = take(spatial_scratch_bytes(n));
    w.total = o;
    return w;
}
size_t oc_spatial_ws_bytes(int64_t n, int dim) {
    if (n < 1 || n > 0x7fffffff || dim < 1 || dim > 16) return 0;
    return oc_spatial_layout(n, dim).total;
}
static OcSpatial oc_spatial_view(const OcSpatialWs &w, void *ws) {
    char *b = static_cast<char *>(ws);
    OcSpatial sp{};
    sp.xs = reinterpret_cast<const float *>(b + w.xs);
    sp.sidx = reinterpret_cast<const int32_t *>(b + w.sidx);
    sp.box = reinterpret_cast<const float *>(b + w.box);
    sp.hq = reinterpret_cast<const float *>(b + w.hq);
    sp.hpid = reinterpret_cast<const long long *>(b + w.hpid);
    sp.hcap = reinterpret_cast<const int32_t *>(b + w.hcap);
    sp.hcapd2 = reinterpret_cast<const float *>(b + w.hcapd2);
    sp.cx = reinterpret_cast<const float *>(b + w.cx);
    sp.cq = reinterpret_cast<const float *>(b + w.cq);
    sp.cpid = reinterpret_cast<const long long *>(b + w.cpid);
    sp.gorder = reinterpret_cast<const uint32_t *>(b + w.gvals_b);
    sp.gkeys = reinterpret_cast<const uint32_t *>(b + w.gkeys_b);
    sp.gstart = reinterpret_cast<const int32_t *>(b + w.gstart);
    sp.n_chunks = w.n_chunks;
    return sp;
}

int oc_forward_spatial_launch(const gnntrk_oc_args *a, float *out, void *ws, size_t ws_bytes, hipStream_t stream) {
    int rc = oc_check(a);
    if (rc) return rc;
    if (a->dim > 16) return fail(GNNTRK_EUNSUPPORTED, "oc_forward_spatial: dim > 16 (use gnntrk_oc_forward)");
    const OcSpatialWs w = oc_spatial_layout(a->n, a->dim);
    if (!out || !ws || ws_bytes < w.total) return fail(GNNTRK_EINVAL, "oc_forward_spatial: workspace too small");
    const OcParams p = oc_params(a);
    char *b = static_cast<char *>(ws);
    rc = spatial_chunks_build(a->x, a->n, a->dim, a->stride, nullptr, 0, reinterpret_cast<float *>(b + w.xs),
                              reinterpret_cast<int32_t *>(b + w.sidx), reinterpret_cast<float *>(b + w.box),
                              b + w.scratch, w.total - w.scratch, stream);
    if (rc) return rc;
    const OcSpatial sp = oc_spatial_view(w, ws);
    const int rgrid = oc_grid((int64_t)w.n_chunks * 64), kgrid = oc_grid(a->n), hgrid = w.n_chunks;
    double *part = reinterpret_cast<double *>(b + w.part);
#define CALL_FS(DP)                                                                                              \
    hipLaunchKernelGGL(oc_hit_records_kernel<DP>, dim3(rgrid), dim3(kOcTpb), 0, stream, p, sp,                    \
                       reinterpret_cast<float *>(b + w.hq), reinterpret_cast<long long *>(b + w.hpid),            \
                       reinterpret_cast<int32_t *>(b + w.hcap), reinterpret_cast<float *>(b + w.hcapd2));         \
    hipLaunchKernelGGL(oc_cp_records_kernel<DP>, dim3(kgrid), dim3(kOcTpb), 0, stream, p,                         \
                       reinterpret_cast<float *>(b + w.cx), reinterpret_cast<float *>(b + w.cq),                  \
                       reinterpret_cast<long long *>(b + w.cpid));                                                \
    hipLaunchKernelGGL((oc_hits_spatial_kernel<DP, false>), dim3(hgrid), dim3(kOcTpb), 0, stream, p, sp,          \
                       (const float *)nullptr, (const float *)nullptr, part, (float *)nullptr, (float *)nullptr)
    if (w.dp == 4) { CALL_FS(4); } else if (w.dp == 8) { CALL_FS(8); } else { CALL_FS(16); }
#undef CALL_FS
    hipLaunchKernelGGL(oc_finalize_spatial_kernel, dim3(1), dim3(kOcTpb), 0, stream, p, (const double *)part, w.n_chunks, out);
    return check_launch("oc_forward_spatial");
}

int oc_backward_spatial_launch(const gnntrk_oc_args *a, const float *g, const float *fwd, float *gx, float *gbeta,
                               int64_t max_cps, void *ws, size_t ws_bytes, hipStream_t stream) {
    int rc = oc_check(a);
    if (rc) return rc;
    if (a->dim > 16) return fail(GNNTRK_EUNSUPPORTED, "oc_backward_spatial: dim > 16 (use gnntrk_oc_backward)");
    if (!g || !fwd || !gx || !gbeta || max_cps < 1) return fail(GNNTRK_EINVAL, "oc_backward_spatial: bad argument");
    const OcSpatialWs w = oc_spatial_layout(a->n, a->dim);
    if (!ws || ws_bytes < w.total) return fail(GNNTRK_EINVAL, "oc_backward_spatial: workspace too small");
    const OcParams p = oc_params(a);
    const OcSpatial sp = oc_spatial_view(w, ws);
    char *b = static_cast<char *>(ws);
    const int hgrid = w.n_chunks, ngrid = oc_grid(a->n);
    // the by-gid ordering of the hits (for the points' attractive share)
    uint32_t *gka = reinterpret_cast<uint32_t *>(b + w.gkeys_a), *gkb = reinterpret_cast<uint32_t *>(b + w.gkeys_b);
    uint32_t *gva = reinterpret_cast<uint32_t *>(b + w.gvals_a), *gvb = reinterpret_cast<uint32_t *>(b + w.gvals_b);
    hipLaunchKernelGGL(oc_gid_keys_kernel, dim3(ngrid), dim3(kOcTpb), 0, stream, a->gid, a->n, gka, gva);
    rc = sort_pairs_u32(gka, gkb, gva, gvb, a->n, 32, b + w.gtemp, sort_pairs_temp_bytes(a->n), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(oc_gid_starts_kernel, dim3(ngrid), dim3(kOcTpb), 0, stream, (const uint32_t *)gkb, a->n,
                       reinterpret_cast<int32_t *>(b + w.gstart));
    const int cgrid = (int)(max_cps < a->n ? max_cps : a->n);
#define CALL_BS(DP)                                                                                              \
    hipLaunchKernelGGL((oc_hits_spatial_kernel<DP, true>), dim3(hgrid), dim3(kOcTpb), 0, stream, p, sp, g, fwd,   \
                       (double *)nullptr, gx, gbeta);                                                             \
    hipLaunchKernelGGL(oc_cps_spatial_kernel<DP>, dim3(cgrid), dim3(kOcTpb), 0, stream, p, sp, g, fwd, gx, gbeta)
    if (w.dp == 4) { CALL_BS(4); } else if (w.dp == 8) { CALL_BS(8); } else { CALL_BS(16); }
#undef CALL_BS
    return check_launch("oc_backward_spatial");
}

}  // namespace gnntrk
