// mickey_amd -- the differentiable tail of the training-time RANSAC, forward AND backward, on gfx950.
// reference lib/models/MicKey/modules/loss/loss_class.py:187-246 (final masked Procrustes of every hypothesis, soft inlier
// score, VCRE / pose loss), loss/solvers.py:13-26,45-52 (weighted_procrustes), utils/training_utils.py:55-61
// (soft_inlier_counting_3d), loss/loss_utils.py:27-69,95-121, lib/utils/metrics.py:56-80 (VCRE on the 7 x 4 x 7 eye grid).
//
// The reference runs this part through ATen + autograd: B*IT_MATCHES*IT_RANSAC (3200) hypotheses x 512 matches tiled into
// [3200, 512, 3] tensors, torch.svd forward and backward (rocSOLVER / MAGMA), ~40 element-wise launches: 6.8 of the 7.7 ms
// of a loss evaluation in round 2.  Here one wave owns a hypothesis:
//   forward   mk_train_tail_fwd      weighted centroids and covariance (fp64 wave sums), 3x3 SVD by one-sided Jacobi (fp64),
//                                    R = V diag(1,1,det) U^T, t, soft inlier score over all S matches, loss; keeps U, S, V, the
//                                    centroids for the backward pass
//   backward  mk_train_tail_bwd_hyp  dL/dR, dL/dt of the loss and score terms, then the CLOSED-FORM adjoint of the Kabsch map
//                                    H -> R (below) -> dL/dH, dL/d(centroids) per hypothesis
//             mk_train_tail_bwd_pts  dL/dX, dL/dY of every match: one thread per (set, match) sums the contributions of the
//                                    set's IT_RANSAC hypotheses in hypothesis order (deterministic, no atomics)
// Adjoint of Kabsch.  H = U S V^T, R = V Z U^T, Z = diag(1, 1, d), d = sign det(U V^T).  For a perturbation dH let
// P = U^T dH V.  The SVD differentials dU = U W_U, dV = V W_V (W antisymmetric) give dR = V K U^T with
//     K_ij = -(P_ij - P_ji) / (s_i + s_j)            z_i = z_j
//     K_ij =  z_j (P_ij + P_ji) / (s_j - s_i)        z_i != z_j   (only pairs with index 3 when d = -1)
// (the 1 / (s_i^2 - s_j^2) poles of the separate U and V differentials cancel: unlike torch.svd's backward this stays finite
// for equal singular values).  With Q = V^T (dL/dR) U:  dL/dH = U C V^T,
//     C_ij = -(Q_ij - Q_ji) / (s_i + s_j)   resp.   C_i3 = C_3i = (Q_i3 + Q_3i) / (s_i - s_3)  when d = -1.
#include "mk_common.hpp"

namespace {
using namespace mk;

constexpr int TT_SLOTS = 16;   // matches per lane: S <= 1024
constexpr int SV = 32;         // floats saved per hypothesis: U 9 | V 9 | S 3 | d 1 | abar 3 | bbar 3 | sum_w 1 | pad 3

struct TailParams {
  const float* X;        // [nsets, S, 3]
  const float* Y;
  const float* mask;     // [nsets * itr, S]  0/1 weights of the differentiable Procrustes
  const float* Rgt;      // [B, 9]
  const float* tgt;      // [B, 3]
  const float* K0;       // [B, 9]  intrinsics of the ORIGINAL images (VCRE only)
  const float* K1;
  int nsets, itr, S, it_matches;
  float th;              // INLIER_3D_TH of the soft score
  int loss_type;         // 0 VCRE, 1 POSE_ERR
  int soft_clip;
  float img_h;           // clip bound of the projected eye points (720, metrics.py:70)
};

// ---- small fp64 3x3 helpers (row-major) ------------------------------------------------------------------------------
__device__ __forceinline__ double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// H = U diag(S) V^T, S descending, by one-sided Jacobi on the columns of H (G = H V has orthogonal columns = U S).
// A vanishing third singular direction is completed by the cross product (sign from the column itself when it is resolvable).
__device__ void svd3(const double* Hin, double* U, double* S, double* V) {
  double G[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 9; ++i) G[i] = Hin[i];
  for (int sweep = 0; sweep < 15; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      double al = 0, be = 0, ga = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        al += G[r * 3 + p] * G[r * 3 + p];
        be += G[r * 3 + q] * G[r * 3 + q];
        ga += G[r * 3 + p] * G[r * 3 + q];
      }
      if (fabs(ga) <= 1e-300 || ga * ga <= 1e-32 * al * be) continue;
      off += fabs(ga);
      const double zeta = (be - al) / (2.0 * ga);
      const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double gp = G[r * 3 + p], gq = G[r * 3 + q];
        G[r * 3 + p] = cs * gp - sn * gq;
        G[r * 3 + q] = sn * gp + cs * gq;
        const double vp = W[r * 3 + p], vq = W[r * 3 + q];
        W[r * 3 + p] = cs * vp - sn * vq;
        W[r * 3 + q] = sn * vp + cs * vq;
      }
    }
    if (off == 0.0) break;
  }
  double nrm[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) nrm[c] = G[c] * G[c] + G[3 + c] * G[3 + c] + G[6 + c] * G[6 + c];
  int o0 = 0, o1 = 1, o2 = 2;
  if (nrm[o0] < nrm[o1]) { const int t = o0; o0 = o1; o1 = t; }
  if (nrm[o0] < nrm[o2]) { const int t = o0; o0 = o2; o2 = t; }
  if (nrm[o1] < nrm[o2]) { const int t = o1; o1 = o2; o2 = t; }
  const int ord[3] = {o0, o1, o2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    S[c] = sqrt(nrm[ord[c]]);
#pragma unroll
    for (int r = 0; r < 3; ++r) V[r * 3 + c] = W[r * 3 + ord[c]];
  }
  // U columns: normalised G columns; re-orthogonalised (Gram-Schmidt) so that tiny singular values do not leak noise
  double u0[3], u1[3], u2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) u0[r] = S[0] > 1e-300 ? G[r * 3 + o0] / S[0] : (r == 0 ? 1.0 : 0.0);
  double d01 = 0.0;
#pragma unroll
  for (int r = 0; r < 3; ++r) { u1[r] = G[r * 3 + o1]; d01 += u0[r] * u1[r]; }
#pragma unroll
  for (int r = 0; r < 3; ++r) u1[r] -= d01 * u0[r];
  double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  if (!(n1 > 1e-12 * S[0]) || !(n1 > 1e-300)) {   // rank 1: any perpendicular
    const int ax = fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2]) ? 0 : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[ax] = 1.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) u1[r] = e[r] - u0[ax] * u0[r];
    n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) u1[r] /= n1;
  u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
  u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
  u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
  const double dir = u2[0] * G[o2] + u2[1] * G[3 + o2] + u2[2] * G[6 + o2];   // the third column itself, when it is not noise
  if (dir < 0.0 && S[2] > 1e-12 * S[0]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) u2[r] = -u2[r];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    U[r * 3 + 0] = u0[r];
    U[r * 3 + 1] = u1[r];
    U[r * 3 + 2] = u2[r];
  }
}

// ---- the eye grid of the VCRE (lib/benchmarks/reprojection.py:34-58): point p of 196 ------------------------------------
__device__ __forceinline__ void eye_point(int p, float* e) {
  const int k = p % 7, i = (p / 7) % 7, j = p / 49;
  e[0] = ((float)i - 3.0f) * 0.3f;
  e[1] = ((float)j - 1.5f) * 0.3f;
  e[2] = (float)k * 0.3f + 1.8f;
}
__device__ __forceinline__ void proj2(const float* K, const float* P, float* uv, float* q) {
#pragma unroll
  for (int a = 0; a < 3; ++a) q[a] = K[a * 3] * P[0] + K[a * 3 + 1] * P[1] + K[a * 3 + 2] * P[2];
  const float iz = 1.0f / (q[2] + 1e-16f);
  uv[0] = q[0] * iz;
  uv[1] = q[1] * iz;
}
__device__ __forceinline__ float clipf(float v, float hi) { return fminf(fmaxf(v, 0.f), hi); }

// vcre_loss(R, t, Rg, tg, K) of metrics.py:56-80 for one hypothesis, the 196 points spread over the wave.  With GRAD the
// adjoint w.r.t. R (GR += ...) and t (Gt += ...) of `scale * vcre` is accumulated (per-lane partial sums: the caller reduces).
template <bool GRAD>
__device__ __forceinline__ float vcre_wave(const float* R, const float* t, const float* Rg, const float* tg, const float* K,
                                           float img_h, int lane, float scale, float* GR, float* Gt) {
  float acc = 0.f;
  for (int p = lane; p < 196; p += 64) {
    float E[3], uvg[2], uv[2], q[3], mv[3], rs[3];
    eye_point(p, E);
    proj2(K, E, uvg, q);
#pragma unroll
    for (int a = 0; a < 3; ++a) mv[a] = R[a * 3] * E[0] + R[a * 3 + 1] * E[1] + R[a * 3 + 2] * E[2] + t[a] - tg[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) rs[a] = Rg[a] * mv[0] + Rg[3 + a] * mv[1] + Rg[6 + a] * mv[2];   // Rg^T (moved - tg)
    proj2(K, rs, uv, q);
    const float ug = clipf(uvg[0], img_h), vg = clipf(uvg[1], img_h);
    const float u = clipf(uv[0], img_h), v = clipf(uv[1], img_h);
    const float du = ug - u, dv = vg - v;
    const float err = sqrtf(du * du + dv * dv + 1e-6f);
    acc += err;
    if (GRAD) {
      // d err / d(u, v) = -(du, dv) / err, through the clip only where the raw coordinate lies inside [0, img_h]
      const float ge = scale * (1.0f / 196.0f) / err;
      float gu = (uv[0] >= 0.f && uv[0] <= img_h) ? -du * ge : 0.f;
      float gv = (uv[1] >= 0.f && uv[1] <= img_h) ? -dv * ge : 0.f;
      const float iz = 1.0f / (q[2] + 1e-16f);
      const float gq[3] = {gu * iz, gv * iz, -(gu * q[0] + gv * q[1]) * iz * iz};
      float grs[3], gmv[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) grs[a] = K[a] * gq[0] + K[3 + a] * gq[1] + K[6 + a] * gq[2];   // K^T gq
#pragma unroll
      for (int a = 0; a < 3; ++a) gmv[a] = Rg[a * 3] * grs[0] + Rg[a * 3 + 1] * grs[1] + Rg[a * 3 + 2] * grs[2];   // Rg grs
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        Gt[a] += gmv[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) GR[a * 3 + b] += gmv[a] * E[b];
      }
    }
  }
  return wave_sum(acc) * (1.0f / 196.0f);
}

// loss value (and, with GRAD, its adjoint w.r.t. R, t scaled by g) of one hypothesis; every lane returns the same values
template <bool GRAD>
__device__ __forceinline__ void loss_wave(const TailParams& p, int b, const float* R, const float* t, int lane, float g,
                                          float* loss, float* lrot, float* ltr, float* GR, float* Gt) {
  float Rg[9], tg[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rg[i] = p.Rgt[(long long)b * 9 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) tg[i] = p.tgt[(long long)b * 3 + i];
  float tr = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) tr += R[i] * Rg[i];
  const float craw = (tr - 1.0f) * 0.5f;
  const float c = fminf(fmaxf(craw, -0.99999f), 0.99999f);
  const float ang = fabsf(acosf(c));
  float l1 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) l1 += fabsf(t[i] - tg[i]);
  *lrot = ang;
  *ltr = l1;
  if (p.loss_type == 1) {   // POSE_ERR (loss_utils.py:27-39)
    const float tr_ = tanhf(ang / 0.9f), tt_ = tanhf(l1 / 0.9f);
    *loss = p.soft_clip ? tr_ + tt_ : ang + l1;
    if (GRAD && lane == 0) {
      const float fr = p.soft_clip ? (1.0f - tr_ * tr_) / 0.9f : 1.0f, ft = p.soft_clip ? (1.0f - tt_ * tt_) / 0.9f : 1.0f;
      const float inside = (craw > -0.99999f && craw < 0.99999f) ? 1.0f : 0.0f;
      const float dang = inside * (-1.0f / sqrtf(1.0f - c * c)) * 0.5f;   // d acos(c) / d trace
#pragma unroll
      for (int i = 0; i < 9; ++i) GR[i] += g * fr * dang * Rg[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) Gt[i] += g * ft * (t[i] > tg[i] ? 1.0f : (t[i] < tg[i] ? -1.0f : 0.0f));
    }
    return;
  }
  // VCRE (loss_utils.py:41-69): the pose under K0 and the inverse pose under K1, averaged, tanh(. / 80)
  float K0[9], K1[9], Ri[9], ti[3], Rgi[9], tgi[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) { K0[i] = p.K0[(long long)b * 9 + i]; K1[i] = p.K1[(long long)b * 9 + i]; }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int c2 = 0; c2 < 3; ++c2) { Ri[a * 3 + c2] = R[c2 * 3 + a]; Rgi[a * 3 + c2] = Rg[c2 * 3 + a]; }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    ti[a] = -(Ri[a * 3] * t[0] + Ri[a * 3 + 1] * t[1] + Ri[a * 3 + 2] * t[2]);
    tgi[a] = -(Rgi[a * 3] * tg[0] + Rgi[a * 3 + 1] * tg[1] + Rgi[a * 3 + 2] * tg[2]);
  }
  const float v1 = vcre_wave<false>(Ri, ti, Rgi, tgi, K1, p.img_h, lane, 0.f, nullptr, nullptr);
  const float v0 = vcre_wave<false>(R, t, Rg, tg, K0, p.img_h, lane, 0.f, nullptr, nullptr);
  const float x = 0.5f * (v1 + v0);
  const float th_ = tanhf(x / 80.0f);
  *loss = p.soft_clip ? th_ : x;
  if (GRAD) {
    const float sc = g * 0.5f * (p.soft_clip ? (1.0f - th_ * th_) / 80.0f : 1.0f);
    float GRi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Gti[3] = {0, 0, 0};
    vcre_wave<true>(Ri, ti, Rgi, tgi, K1, p.img_h, lane, sc, GRi, Gti);
    vcre_wave<true>(R, t, Rg, tg, K0, p.img_h, lane, sc, GR, Gt);
    // Ri = R^T, ti = -R^T t:  dL/dR += GRi^T - t (x) Gti,  dL/dt += -R Gti
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int c2 = 0; c2 < 3; ++c2) GR[a * 3 + c2] += GRi[c2 * 3 + a] - t[a] * Gti[c2];
      Gt[a] -= R[a * 3] * Gti[0] + R[a * 3 + 1] * Gti[1] + R[a * 3 + 2] * Gti[2];
    }
  }
}

// ---- forward: one workgroup per match set, one wave per hypothesis ------------------------------------------------------
__global__ __launch_bounds__(256) void tail_fwd_kernel(TailParams p, float* __restrict__ out, float* __restrict__ Rt,
                                                       float* __restrict__ sv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // X[S*3] | Y[S*3]
  float* sX = lds;
  float* sY = lds + (size_t)p.S * 3;
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < p.S * 3; i += 256) {
    sX[i] = p.X[(long long)r * p.S * 3 + i];
    sY[i] = p.Y[(long long)r * p.S * 3 + i];
  }
  __syncthreads();
  const int b = r / p.it_matches;
  for (int h = wave; h < p.itr; h += 4) {
    const long long hyp = (long long)r * p.itr + h;
    const float* w = p.mask + hyp * p.S;
    double sw = 0.0, sa[3] = {0, 0, 0}, sb[3] = {0, 0, 0};
    float wl[TT_SLOTS];
#pragma unroll
    for (int q = 0; q < TT_SLOTS; ++q) {
      const int j = q * 64 + lane;
      wl[q] = j < p.S ? w[j] : 0.f;
      if (wl[q] != 0.f) {
        sw += fabs((double)wl[q]);
#pragma unroll
        for (int a = 0; a < 3; ++a) { sa[a] += (double)wl[q] * sX[j * 3 + a]; sb[a] += (double)wl[q] * sY[j * 3 + a]; }
      }
    }
    sw = wave_sum_d(sw);
    const double inv = 1.0 / (sw + 1e-16);
    double am[3], bm[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { am[a] = wave_sum_d(sa[a]) * inv; bm[a] = wave_sum_d(sb[a]) * inv; }
    double hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < TT_SLOTS; ++q)
      if (wl[q] != 0.f) {
        const int j = q * 64 + lane;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) hl[a * 3 + c] += (double)wl[q] * ((double)sX[j * 3 + a] - am[a]) * ((double)sY[j * 3 + c] - bm[c]);
      }
    double H[9], U[9], Sg[3], V[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = wave_sum_d(hl[i]);
    svd3(H, U, Sg, V);
    const double d = det3(U) * det3(V) >= 0.0 ? 1.0 : -1.0;   // sign det(U V^T)
    float R[9], t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) R[a * 3 + c] = (float)(V[a * 3] * U[c * 3] + V[a * 3 + 1] * U[c * 3 + 1] + d * V[a * 3 + 2] * U[c * 3 + 2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a] = (float)bm[a] - (R[a * 3] * (float)am[0] + R[a * 3 + 1] * (float)am[1] + R[a * 3 + 2] * (float)am[2]);
    // soft inlier score over ALL matches of the set (training_utils.py:55-61)
    const float beta = 5.0f / p.th;
    float sc = 0.f;
#pragma unroll
    for (int q = 0; q < TT_SLOTS; ++q) {
      const int j = q * 64 + lane;
      if (j < p.S) {
        float e2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float e = R[a * 3] * sX[j * 3] + R[a * 3 + 1] * sX[j * 3 + 1] + R[a * 3 + 2] * sX[j * 3 + 2] + t[a] - sY[j * 3 + a];
          e2 += e * e;
        }
        sc += 1.0f / (1.0f + __expf(-beta * (p.th - sqrtf(e2 + 1e-6f))));
      }
    }
    sc = wave_sum(sc);
    float loss, lrot, ltr;
    loss_wave<false>(p, b, R, t, lane, 0.f, &loss, &lrot, &ltr, nullptr, nullptr);
    if (lane == 0) {
      out[hyp * 4 + 0] = loss; out[hyp * 4 + 1] = lrot; out[hyp * 4 + 2] = ltr; out[hyp * 4 + 3] = sc;
#pragma unroll
      for (int i = 0; i < 9; ++i) Rt[hyp * 12 + i] = R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) Rt[hyp * 12 + 9 + i] = t[i];
      float* s = sv + hyp * SV;
#pragma unroll
      for (int i = 0; i < 9; ++i) { s[i] = (float)U[i]; s[9 + i] = (float)V[i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { s[18 + i] = (float)Sg[i]; s[22 + i] = (float)am[i]; s[25 + i] = (float)bm[i]; }
      s[21] = (float)d;
      s[28] = (float)sw;
    }
  }
}

// ---- backward, per hypothesis: g = (dL/d loss_value, dL/d score) -> G_H (9), d abar (3), d bbar (3) -----------------------
__global__ __launch_bounds__(256) void tail_bwd_hyp_kernel(TailParams p, const float* __restrict__ Rt, const float* __restrict__ sv,
                                                           const float* __restrict__ g, float* __restrict__ hv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sX = lds;
  float* sY = lds + (size_t)p.S * 3;
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < p.S * 3; i += 256) {
    sX[i] = p.X[(long long)r * p.S * 3 + i];
    sY[i] = p.Y[(long long)r * p.S * 3 + i];
  }
  __syncthreads();
  const int b = r / p.it_matches;
  for (int h = wave; h < p.itr; h += 4) {
    const long long hyp = (long long)r * p.itr + h;
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rt[hyp * 12 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = Rt[hyp * 12 + 9 + i];
    const float gl = g[hyp * 2 + 0], gs = g[hyp * 2 + 1];
    float GR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Gt[3] = {0, 0, 0};
    // score term: s = sum_j sigmoid(beta (th - d_j)),  d_j = sqrt(|e_j|^2 + 1e-6),  e_j = R x_j + t - y_j
    const float beta = 5.0f / p.th;
#pragma unroll
    for (int q = 0; q < TT_SLOTS; ++q) {
      const int j = q * 64 + lane;
      if (j < p.S) {
        float e[3], e2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          e[a] = R[a * 3] * sX[j * 3] + R[a * 3 + 1] * sX[j * 3 + 1] + R[a * 3 + 2] * sX[j * 3 + 2] + t[a] - sY[j * 3 + a];
          e2 += e[a] * e[a];
        }
        const float dj = sqrtf(e2 + 1e-6f);
        const float sg = 1.0f / (1.0f + __expf(-beta * (p.th - dj)));
        const float f = gs * (-beta) * sg * (1.0f - sg) / dj;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          Gt[a] += f * e[a];
#pragma unroll
          for (int c = 0; c < 3; ++c) GR[a * 3 + c] += f * e[a] * sX[j * 3 + c];
        }
      }
    }
    float loss, lrot, ltr;
    loss_wave<true>(p, b, R, t, lane, gl, &loss, &lrot, &ltr, GR, Gt);
#pragma unroll
    for (int i = 0; i < 9; ++i) GR[i] = wave_sum(GR[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) Gt[i] = wave_sum(Gt[i]);
    if (lane == 0) {
      const float* s = sv + hyp * SV;
      double U[9], V[9], Sg[3], am[3], bm[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) { U[i] = s[i]; V[i] = s[9 + i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { Sg[i] = s[18 + i]; am[i] = s[22 + i]; bm[i] = s[25 + i]; }
      const double d = s[21];
      // t = bbar - R abar:  dL/dR += -Gt (x) abar,  dL/d abar = -R^T Gt,  dL/d bbar = Gt
      double GRd[9], da[3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) GRd[a * 3 + c] = (double)GR[a * 3 + c] - (double)Gt[a] * am[c];
#pragma unroll
      for (int a = 0; a < 3; ++a) da[a] = -((double)R[a] * Gt[0] + (double)R[3 + a] * Gt[1] + (double)R[6 + a] * Gt[2]);
      // Q = V^T GR U;  C from the closed-form adjoint;  G_H = U C V^T
      double Tm[9], Q[9], C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Tm[a * 3 + c] = V[a] * GRd[c] + V[3 + a] * GRd[3 + c] + V[6 + a] * GRd[6 + c];   // V^T GR
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Q[a * 3 + c] = Tm[a * 3] * U[c] + Tm[a * 3 + 1] * U[3 + c] + Tm[a * 3 + 2] * U[6 + c];
      const double z[3] = {1.0, 1.0, d};
      const double tiny = 1e-12 * (Sg[0] > 0 ? Sg[0] : 1.0);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j2 = 0; j2 < 3; ++j2) {
          if (i == j2) continue;
          if (z[i] == z[j2]) {
            const double den = Sg[i] + Sg[j2];
            C[i * 3 + j2] = den > tiny ? -(Q[i * 3 + j2] - Q[j2 * 3 + i]) / den : 0.0;
          } else {   // reflection case: one of the two is index 2 with z = -1
            const int o = i == 2 ? j2 : i;                 // the index with z = +1
            const double den = Sg[o] - Sg[2];
            C[i * 3 + j2] = fabs(den) > tiny ? (Q[i * 3 + j2] + Q[j2 * 3 + i]) / den : 0.0;
          }
        }
      double UC[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) UC[a * 3 + c] = U[a * 3] * C[c] + U[a * 3 + 1] * C[3 + c] + U[a * 3 + 2] * C[6 + c];
      float* o = hv + hyp * 16;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[a * 3 + c] = (float)(UC[a * 3] * V[c * 3] + UC[a * 3 + 1] * V[c * 3 + 1] + UC[a * 3 + 2] * V[c * 3 + 2]);
#pragma unroll
      for (int a = 0; a < 3; ++a) { o[9 + a] = (float)da[a]; o[12 + a] = Gt[a]; }
      o[15] = gs;
    }
  }
}

// ---- backward, per match: dL/dX[r, j], dL/dY[r, j] = sum over the set's hypotheses, in hypothesis order --------------------
__global__ __launch_bounds__(256) void tail_bwd_pts_kernel(TailParams p, const float* __restrict__ Rt, const float* __restrict__ sv,
                                                           const float* __restrict__ hv, float* __restrict__ gX,
                                                           float* __restrict__ gY) {
  const int r = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.S) return;
  float x[3], y[3], ax[3] = {0, 0, 0}, ay[3] = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < 3; ++a) { x[a] = p.X[((long long)r * p.S + j) * 3 + a]; y[a] = p.Y[((long long)r * p.S + j) * 3 + a]; }
  const float beta = 5.0f / p.th;
  for (int h = 0; h < p.itr; ++h) {
    const long long hyp = (long long)r * p.itr + h;
    const float* R = Rt + hyp * 12;      // wave-uniform addresses: scalar loads
    const float* s = sv + hyp * SV;
    const float* o = hv + hyp * 16;
    const float w = p.mask[hyp * p.S + j];
    const float gs = o[15];
    float e[3], e2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      e[a] = R[a * 3] * x[0] + R[a * 3 + 1] * x[1] + R[a * 3 + 2] * x[2] + R[9 + a] - y[a];
      e2 += e[a] * e[a];
    }
    const float dj = sqrtf(e2 + 1e-6f);
    const float sg = 1.0f / (1.0f + __expf(-beta * (p.th - dj)));
    const float f = gs * (-beta) * sg * (1.0f - sg) / dj;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      ax[a] += f * (R[a] * e[0] + R[3 + a] * e[1] + R[6 + a] * e[2]);   // R^T g_e
      ay[a] -= f * e[a];
    }
    if (w != 0.f) {
      const float wn = w / (s[28] + 1e-16f);
      float xc[3], yc[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) { xc[a] = x[a] - s[22 + a]; yc[a] = y[a] - s[25 + a]; }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        ax[a] += w * (o[a * 3] * yc[0] + o[a * 3 + 1] * yc[1] + o[a * 3 + 2] * yc[2]) + wn * o[9 + a];    // G_H (y - bbar) + d abar
        ay[a] += w * (o[a] * xc[0] + o[3 + a] * xc[1] + o[6 + a] * xc[2]) + wn * o[12 + a];               // G_H^T (x - abar) + d bbar
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { gX[((long long)r * p.S + j) * 3 + a] = ax[a]; gY[((long long)r * p.S + j) * 3 + a] = ay[a]; }
}

int fill(TailParams& p, const float* X, const float* Y, const float* mask, const float* Rgt, const float* tgt, const float* K0,
         const float* K1, int nsets, int itr, int S, int it_matches, float th, int loss_type, int soft_clip, float img_h) {
  MK_CHECK_ARG(X && Y && mask && Rgt && tgt, "mk_train_tail: null pointer");
  MK_CHECK_ARG(nsets > 0 && itr > 0 && S > 0 && S <= 64 * TT_SLOTS && it_matches > 0 && nsets % it_matches == 0, "mk_train_tail: bad sizes (S <= %d)", 64 * TT_SLOTS);
  MK_CHECK_ARG(loss_type == 1 || (K0 && K1), "mk_train_tail: the VCRE needs the intrinsics of the original images");
  MK_CHECK_ARG(th > 0.f, "mk_train_tail: the soft-inlier threshold must be positive");
  p = TailParams{X, Y, mask, Rgt, tgt, K0, K1, nsets, itr, S, it_matches, th, loss_type, soft_clip, img_h};
  return MK_OK;
}

}  // namespace

// ---- aggregation: hypotheses of a set -> the set's loss, sets of a pair -> the pair's baseline ---------------------------------
// reference loss_class.py:229-246 (softmax of the scores / temperature over the it_ransac hypotheses of a set, with the null
// hypothesis as one more column; rotation / translation errors under the softmax WITHOUT that column) and :263-268 (sums over the
// it_matches sets of a pair).  One workgroup per pair, one wave per set (lanes = hypotheses), the pair's sums in set order.
// coef [nhyp, 2] = d loss_value(set) / d (loss_k, score_k): the backward is g(pair) * coef.
__global__ __launch_bounds__(256) void train_aggregate_fwd_kernel(const float* __restrict__ out, const float* __restrict__ Rt,
                                                                  const float* __restrict__ saved, int it_r, int it_m, float temp,
                                                                  int add_null, float null_loss, float null_score,
                                                                  float* __restrict__ loss_value, float* __restrict__ per_pair,
                                                                  float* __restrict__ coef, int* __restrict__ flags) {
  __shared__ float srow[3][64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bad = 0, rank1 = 0;
  for (int o = wave; o < it_m; o += 4) {
    const long long row = (long long)b * it_m + o;
    float vmax = add_null ? null_score / temp : -INFINITY;
    for (int h = lane; h < it_r; h += 64) vmax = fmaxf(vmax, out[(row * it_r + h) * 4 + 3] / temp);
    vmax = wave_max(vmax);
    float z = 0.f, zl = 0.f, zr = 0.f, zt = 0.f;
    for (int h = lane; h < it_r; h += 64) {
      const float* q = out + (row * it_r + h) * 4;
      const float e = expf(q[3] / temp - vmax);
      z += e; zl += e * q[0]; zr += e * q[1]; zt += e * q[2];
      const float* rt = Rt + (row * it_r + h) * 12;
#pragma unroll
      for (int i = 0; i < 12; ++i) bad |= !(fabsf(rt[i]) <= 3.0e38f);   // NaN or inf
      const float* sv = saved + (row * it_r + h) * 32 + 18;               // singular values of H: rank one = exactly one above max * 3 eps
      const float smax = fmaxf(sv[0], fmaxf(sv[1], sv[2])), tol = smax * 3.f * 1.1920929e-07f;
      rank1 += ((sv[0] > tol) + (sv[1] > tol) + (sv[2] > tol)) == 1;
    }
    z = wave_sum(z); zl = wave_sum(zl); zr = wave_sum(zr); zt = wave_sum(zt);
    const float en = add_null ? expf(null_score / temp - vmax) : 0.f;
    const float Z = z + en;
    const float lv = (zl + en * null_loss) / Z;
    for (int h = lane; h < it_r; h += 64) {
      const float* q = out + (row * it_r + h) * 4;
      const float sm = expf(q[3] / temp - vmax) / Z;
      coef[(row * it_r + h) * 2 + 0] = sm;
      coef[(row * it_r + h) * 2 + 1] = sm * (q[0] - lv) / temp;
    }
    if (lane == 0) {
      loss_value[row] = lv;
      srow[0][o & 63] = lv; srow[1][o & 63] = zr / z; srow[2][o & 63] = zt / z;
    }
  }
  bad = __any(bad);
  rank1 = (int)wave_sum((float)rank1);
  if (lane == 0) {
    if (bad) atomicOr(flags + 0, 1);
    if (rank1) atomicAdd(flags + 1, rank1);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float a = 0.f;
    for (int o = 0; o < it_m; ++o) a += srow[threadIdx.x][o];
    per_pair[b * 3 + threadIdx.x] = a;
  }
}

__global__ __launch_bounds__(256) void train_aggregate_bwd_kernel(const float* __restrict__ coef, const float* __restrict__ g_pair,
                                                                  int per_pair_hyp, long long nhyp, float* __restrict__ g) {
  const long long h = blockIdx.x * 256LL + threadIdx.x;
  if (h >= nhyp) return;
  const float gp = g_pair[h / per_pair_hyp];
  g[h * 2 + 0] = gp * coef[h * 2 + 0];
  g[h * 2 + 1] = gp * coef[h * 2 + 1];
}

extern "C" {

int mk_train_aggregate_fwd(const float* out, const float* Rt, const float* saved, int B, int it_matches, int it_ransac,
                           float temperature, int add_null, float null_loss, float null_score, float* loss_value, float* per_pair,
                           float* coef, int* flags, mk_stream_t stream) {
  MK_CHECK_ARG(out && Rt && saved && loss_value && per_pair && coef && flags, "mk_train_aggregate_fwd: null pointer");
  MK_CHECK_ARG(B > 0 && it_matches > 0 && it_matches <= 64 && it_ransac > 0 && temperature > 0.f,
               "mk_train_aggregate_fwd: bad sizes (it_matches <= 64)");
  hipLaunchKernelGGL(train_aggregate_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, out, Rt, saved, it_ransac, it_matches,
                     temperature, add_null, null_loss, null_score, loss_value, per_pair, coef, flags);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_train_aggregate_bwd(const float* coef, const float* g_pair, int B, int it_matches, int it_ransac, float* grad_out,
                           mk_stream_t stream) {
  MK_CHECK_ARG(coef && g_pair && grad_out && B > 0 && it_matches > 0 && it_ransac > 0, "mk_train_aggregate_bwd: bad args");
  const long long nh = (long long)B * it_matches * it_ransac;
  hipLaunchKernelGGL(train_aggregate_bwd_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coef, g_pair,
                     it_matches * it_ransac, nh, grad_out);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_train_tail_fwd(const float* X, const float* Y, const float* mask, const float* Rgt, const float* tgt, const float* K0,
                      const float* K1, int nsets, int it_ransac, int S, int it_matches, float th_soft, int loss_type,
                      int soft_clip, float img_h, float* out, float* Rt, float* saved, mk_stream_t stream) {
  TailParams p;
  if (int e = fill(p, X, Y, mask, Rgt, tgt, K0, K1, nsets, it_ransac, S, it_matches, th_soft, loss_type, soft_clip, img_h)) return e;
  MK_CHECK_ARG(out && Rt && saved, "mk_train_tail_fwd: null output");
  hipLaunchKernelGGL(tail_fwd_kernel, dim3(nsets), dim3(256), (size_t)S * 6 * sizeof(float), (hipStream_t)stream, p, out, Rt, saved);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

int mk_train_tail_bwd(const float* X, const float* Y, const float* mask, const float* Rgt, const float* tgt, const float* K0,
                      const float* K1, int nsets, int it_ransac, int S, int it_matches, float th_soft, int loss_type,
                      int soft_clip, float img_h, const float* Rt, const float* saved, const float* grad_out, float* work,
                      float* gX, float* gY, mk_stream_t stream) {
  TailParams p;
  if (int e = fill(p, X, Y, mask, Rgt, tgt, K0, K1, nsets, it_ransac, S, it_matches, th_soft, loss_type, soft_clip, img_h)) return e;
  MK_CHECK_ARG(Rt && saved && grad_out && work && gX && gY, "mk_train_tail_bwd: null pointer");
  hipLaunchKernelGGL(tail_bwd_hyp_kernel, dim3(nsets), dim3(256), (size_t)S * 6 * sizeof(float), (hipStream_t)stream, p, Rt, saved,
                     grad_out, work);
  MK_CHECK_LAUNCH();
  hipLaunchKernelGGL(tail_bwd_pts_kernel, dim3((S + 255) / 256, nsets), dim3(256), 0, (hipStream_t)stream, p, Rt, saved, work, gX, gY);
  MK_CHECK_LAUNCH();
  return MK_OK;
}

}  // extern "C"
