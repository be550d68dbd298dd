// Policy-side kernels of the env-step (round 3): what sits between the library GEMMs of the two policies.
//
//   k_mcp_tail       PolicyMCP's last layer + mixing stage as ONE fp32 MFMA kernel (uhc/core/policy_mcp.py:30-38, policy.py:12-15):
//                      out[n, a] = sum_k softmax(logits[n, :])_k * (b3[k, a] + sum_j relu(h2[k, n, j] + b2[k, j]) W3[k, j, a])  (+ std[a] noise[n, a])
//                    The softmax weight of (n, k) scales row n of primitive k's hidden activations, so the K small GEMMs [n, J] x [J, A] and
//                    the weighted sum over k collapse into one [n, K J] x [K J, A] product whose A operand is formed on the fly (bias, relu,
//                    weight) from the second batched GEMM's raw output.  Replaces: broadcast-bias copy + relu pass + batched GEMM (A = 75 columns:
//                    a bad shape for the library, 63 TFLOP/s) + bias copy + mixing kernel.
//   k_gru_cell_step  torch.nn.GRUCell's gate math after the two gate GEMMs (kin_poly/models/traj_ar_smpl_net.py:333-343 through
//                    uhc/khrylib/models/rnn.py:24-36), writing the new hidden state both as the next step's GEMM operand and straight into the
//                    [state | h] row the action MLP reads (no torch.cat pass, no workspace for a backward that never runs).
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32): lane l feeds A[l & 15][l >> 4] and B[l >> 4][l & 15]; its four results are
// D[4 (l >> 4) + r][l & 15].  A workgroup owns 16 rows (envs); its 8 waves split the K * J products into chunks of 64 and meet in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kp {

typedef float kp_f32x4 __attribute__((ext_vector_type(4)));

constexpr int MCP_WAVES = 8;        // waves per workgroup (two per SIMD: one wave's operand loads under the other's MFMAs)
constexpr int MCP_CHUNK = 64;       // hidden units per work unit

// VEC (NTILE == 5 only): w3 rows are ldw >= 80 floats apart, 16-byte aligned, so a lane takes its columns 4 i .. 4 i + 3 of a row as ONE
// 16-byte load (tiles 0..3 hold columns 4 i + t) and column 64 + i as tile 4: 2 loads per row instead of 5, a quarter of the cache-line
// look-ups.  Otherwise tile t holds columns 16 t + i (dword loads).  Columns >= A of a tile re-read column A - 1 into results nobody stores.
template <int NTILE, bool VEC>      // output columns in tiles of 16: A <= 16 * NTILE
__global__ __launch_bounds__(64 * MCP_WAVES) void k_mcp_tail(int n, int K, int J, int A, const float* __restrict__ h2, const float* __restrict__ b2,
                                                              const float* __restrict__ w3, int ldw, const float* __restrict__ b3, const float* __restrict__ logits,
                                                              const float* __restrict__ noise, int noise_stride, const float* __restrict__ stdv,
                                                              float* __restrict__ out) {
    static_assert(!VEC || NTILE == 5, "the 16-byte operand path is laid out for 80 columns");
    __shared__ float wsm[16][17];                                  // softmax weights of the block's rows (K <= 16)
    __shared__ float red[MCP_WAVES][16][16 * NTILE + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n0 = blockIdx.x * 16;
    if (tid < 16) {
        const int row = n0 + tid;
        if (row < n) {
            const float* lg = logits + (size_t)row * K;
            float mx = lg[0];
            for (int k = 1; k < K; k++) mx = fmaxf(mx, lg[k]);
            float den = 0.f;
            for (int k = 0; k < K; k++) { const float w = expf(lg[k] - mx); wsm[tid][k] = w; den += w; }
            const float inv = 1.0f / den;
            for (int k = 0; k < K; k++) wsm[tid][k] *= inv;
        } else {
            for (int k = 0; k < K; k++) wsm[tid][k] = 0.f;
        }
    }
    __syncthreads();
    const int i = lane & 15, q = lane >> 4;
    const int rowA = min(n0 + i, n - 1);                            // rows past the end repeat the last one; their results are never stored
    kp_f32x4 acc[NTILE];
    int ct[NTILE];                                                  // this lane's column in every tile
#pragma unroll
    for (int t = 0; t < NTILE; t++) {
        acc[t] = kp_f32x4{0.f, 0.f, 0.f, 0.f};
        ct[t] = VEC ? (t < 4 ? 4 * i + t : 64 + i) : min(16 * t + i, A - 1);
    }
    const int cpk = J / MCP_CHUNK, units = K * cpk;
    constexpr int SPU = MCP_CHUNK / 16;                             // steps of 16 hidden units per work unit
    const int steps = ((units - wv + MCP_WAVES - 1) / MCP_WAVES) * SPU;
    // operands of step g: 16 hidden units of one primitive = four MFMA k-slices per column tile
    auto fetch = [&](int g, float4& hv, float4& bv, float& sc, float (&bw)[4][NTILE]) {
        const int u = wv + MCP_WAVES * (g / SPU), k = u / cpk, jb = (u - k * cpk) * MCP_CHUNK + 16 * (g % SPU) + 4 * q;
        hv = *reinterpret_cast<const float4*>(h2 + ((size_t)k * n + rowA) * J + jb);
        bv = *reinterpret_cast<const float4*>(b2 + (size_t)k * J + jb);
        sc = wsm[i][k];
        const float* wk = w3 + ((size_t)k * J + jb) * ldw;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (VEC) {
                const float4 v = *reinterpret_cast<const float4*>(wk + (size_t)m * ldw + 4 * i);
                bw[m][0] = v.x; bw[m][1] = v.y; bw[m][2] = v.z; bw[m][3] = v.w;
                bw[m][NTILE - 1] = wk[(size_t)m * ldw + 64 + i];
            } else {
#pragma unroll
                for (int t = 0; t < NTILE; t++) bw[m][t] = wk[(size_t)m * ldw + ct[t]];
            }
        }
    };
    auto multiply = [&](const float4& hv, const float4& bv, float sc, const float (&bw)[4][NTILE]) {
        const float av[4] = {fmaxf(hv.x + bv.x, 0.f) * sc, fmaxf(hv.y + bv.y, 0.f) * sc, fmaxf(hv.z + bv.z, 0.f) * sc, fmaxf(hv.w + bv.w, 0.f) * sc};
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int t = 0; t < NTILE; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bw[m][t], acc[t], 0, 0, 0);
    };
    // two operand sets in ping-pong (steps is a multiple of SPU = 4): the loads of one are in flight while the other multiplies; the
    // scheduling barriers keep the compiler from sinking the loads below the MFMA block they are meant to hide under
    static_assert(SPU % 2 == 0, "ping-pong over pairs of steps");
    float4 h0, c0, h1, c1; float s0, s1; float w0[4][NTILE], w1[4][NTILE];
    if (steps > 0) fetch(0, h0, c0, s0, w0);
    for (int g = 0; g < steps; g += 2) {
        fetch(g + 1, h1, c1, s1, w1);
        __builtin_amdgcn_sched_barrier(0);
        multiply(h0, c0, s0, w0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(min(g + 2, steps - 1), h0, c0, s0, w0);               // the last pair re-reads its own second step
        __builtin_amdgcn_sched_barrier(0);
        multiply(h1, c1, s1, w1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NTILE; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wv][4 * q + r][VEC ? ct[t] : 16 * t + i] = acc[t][r];
    __syncthreads();
    for (int idx = tid; idx < 16 * A; idx += 64 * MCP_WAVES) {
        const int row = idx / A, a = idx - row * A;
        if (n0 + row >= n) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < MCP_WAVES; w++) v += red[w][row][a];
        for (int k = 0; k < K; k++) v += wsm[row][k] * b3[(size_t)k * A + a];
        if (noise) v += stdv[a] * noise[(size_t)(n0 + row) * noise_stride + a];
        out[(size_t)(n0 + row) * A + a] = v;
    }
}

__device__ __forceinline__ float kp_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// gi = x W_ih^T, gh = h W_hh^T WITHOUT their biases (the plain GEMMs whose library solutions are pinned in assets/tunableop_gfx950.csv);
// r = sigma(gi_r + b_ir + gh_r + b_hr), z likewise, n = tanh(gi_n + b_in + r (gh_n + b_hn)), h' = (1 - z) n + z h        (torch.nn.GRUCell)
// h_out [n, H] (may alias h_in); xcat [n, D + H] (optional) <- [state | h'].
__global__ void k_gru_cell_step(int n, int H, int D, const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ b_ih,
                                const float* __restrict__ b_hh, const float* h_in, const float* __restrict__ state, float* h_out, float* __restrict__ xcat) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * H) return;
    const int e = (int)(idx / H), j = (int)(idx - (size_t)e * H);
    const float* a = gi + (size_t)e * 3 * H; const float* b = gh + (size_t)e * 3 * H;
    const float r = kp_sigmoid(a[j] + b_ih[j] + b[j] + b_hh[j]);
    const float z = kp_sigmoid(a[H + j] + b_ih[H + j] + b[H + j] + b_hh[H + j]);
    const float nn = tanhf(a[2 * H + j] + b_ih[2 * H + j] + r * (b[2 * H + j] + b_hh[2 * H + j]));
    const float h = (1.0f - z) * nn + z * h_in[idx];
    h_out[idx] = h;
    if (xcat) {
        float* xr = xcat + (size_t)e * (D + H);
        xr[D + j] = h;
        if (j < D) xr[j] = state[(size_t)e * D + j];
    }
}

}  // namespace kp
