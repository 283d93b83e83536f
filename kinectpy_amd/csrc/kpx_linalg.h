// kpx_linalg.h -- tiny dense fp64 linear algebra for single-thread device use.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace kpx {

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 given as a = (xx,xy,xz,yy,yz,zz).
// On return w[0] <= w[1] <= w[2] and V[:, k] (column k, V row-major 3x3) is the k-th eigenvector;
// det(V) = +1 before the sort, columns are swapped (with one sign flip per swap) to keep det = +1.
__host__ __device__ __forceinline__ void sym3_eigen(const double a[6], double w[3], double V[9])
{
    double A[3][3] = { { a[0], a[1], a[2] }, { a[1], a[3], a[4] }, { a[2], a[4], a[5] } };
    double Q[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag || off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double apq = A[p][q];
            if (apq == 0.0) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {           // A <- A J
                double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {           // A <- J^T A
                double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {           // Q <- Q J
                double qkp = Q[k][p], qkq = Q[k][q];
                Q[k][p] = c * qkp - s * qkq;
                Q[k][q] = s * qkp + c * qkq;
            }
        }
    }
    double e[3] = { A[0][0], A[1][1], A[2][2] };
    // sort ascending by swapping columns; flip one column's sign per swap to keep det(Q) = +1
#define KPX_SWAPCOL(i, j)                                                                          \
    if (e[i] > e[j]) {                                                                             \
        double te = e[i]; e[i] = e[j]; e[j] = te;                                                  \
        for (int k = 0; k < 3; ++k) { double tq = Q[k][i]; Q[k][i] = Q[k][j]; Q[k][j] = -tq; }     \
    }
    KPX_SWAPCOL(0, 1) KPX_SWAPCOL(1, 2) KPX_SWAPCOL(0, 1)
#undef KPX_SWAPCOL
    for (int k = 0; k < 3; ++k) { w[k] = e[k]; for (int r = 0; r < 3; ++r) V[3 * r + k] = Q[r][k]; }
}

// Rotation of the Umeyama/Kabsch solution for covariance S (row-major 3x3, S = E[t s^T] - mu_t mu_s^T):
// R = U diag(1,1,sign(det U det V)) V^T for S = U D V^T.  Built from the eigenvectors of S^T S
// (V proper, descending), U columns u1,u2 = normalised S v1, S v2 (Gram-Schmidt), u3 = u1 x u2:
// algebraically identical to the signed-SVD form and well defined when the smallest singular
// value vanishes (planar data).
__host__ __device__ __forceinline__ void kabsch_rotation(const double S[9], double R[9])
{
    double M[6];     // S^T S
    M[0] = S[0] * S[0] + S[3] * S[3] + S[6] * S[6];
    M[1] = S[0] * S[1] + S[3] * S[4] + S[6] * S[7];
    M[2] = S[0] * S[2] + S[3] * S[5] + S[6] * S[8];
    M[3] = S[1] * S[1] + S[4] * S[4] + S[7] * S[7];
    M[4] = S[1] * S[2] + S[4] * S[5] + S[7] * S[8];
    M[5] = S[2] * S[2] + S[5] * S[5] + S[8] * S[8];
    double w[3], V[9];
    sym3_eigen(M, w, V);
    // descending order: v1 = col 2, v2 = col 1, v3 = v1 x v2 (keeps V proper)
    double v1[3] = { V[2], V[5], V[8] }, v2[3] = { V[1], V[4], V[7] };
    double v3[3] = { v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0] };
    double u1[3], u2[3];
    for (int r = 0; r < 3; ++r) {
        u1[r] = S[3 * r] * v1[0] + S[3 * r + 1] * v1[1] + S[3 * r + 2] * v1[2];
        u2[r] = S[3 * r] * v2[0] + S[3 * r + 1] * v2[1] + S[3 * r + 2] * v2[2];
    }
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    if (!(n1 > 0.0)) { for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0; return; }
    for (int r = 0; r < 3; ++r) u1[r] /= n1;
    double dp = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int r = 0; r < 3; ++r) u2[r] -= dp * u1[r];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (!(n2 > 1e-300)) {
        // rank-1 covariance: any u2 orthogonal to u1 (degenerate input; Open3D's result is arbitrary too)
        double ax[3] = { fabs(u1[0]) < 0.9 ? 1.0 : 0.0, fabs(u1[0]) < 0.9 ? 0.0 : 1.0, 0.0 };
        u2[0] = u1[1] * ax[2] - u1[2] * ax[1]; u2[1] = u1[2] * ax[0] - u1[0] * ax[2]; u2[2] = u1[0] * ax[1] - u1[1] * ax[0];
        n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    }
    for (int r = 0; r < 3; ++r) u2[r] /= n2;
    double u3[3] = { u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0] };
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[3 * r + c] = u1[r] * v1[c] + u2[r] * v2[c] + u3[r] * v3[c];
}

// Solve the symmetric positive (semi-)definite 6x6 system A x = b by LDL^T without pivoting
// (Open3D: JTJ.ldlt().solve(-JTr)).  A row-major, destroyed.  Returns false if a pivot is ~0.
__host__ __device__ __forceinline__ bool solve6_ldlt(double A[36], const double b[6], double x[6])
{
    double L[36], D[6];
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        D[j] = d;
        if (!(fabs(d) > 1e-300)) return false;
        L[6 * j + j] = 1.0;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[6 * i + j];
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = v / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k]; y[i] = v; }
    for (int i = 0; i < 6; ++i) y[i] /= D[i];
    for (int i = 5; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k]; x[i] = v; }
    return true;
}

}  // namespace kpx
