/*
 * o3d_oracle.c -- CPU ORACLE (test infrastructure only; Open3D parts PARITY UNPINNED, open3d_slam's own parts pinned: see o3d_oracle.h).
 *
 * Restates, in plain C + OpenMP, the algorithm of the reference's scan-to-map
 * point-to-plane ICP / map-fusion hot path.  In-tree citations are relative to
 * /root/reference/open3d_slam/open3d_slam/; "[O3D]" marks behaviour of Open3D
 * v0.15.1 (third-party, absent from the tree) restated from its published
 * algorithm as summarised in SURVEY.md Appendix A.
 */
#include "o3d_oracle.h"

#include <float.h>
#include <math.h>
#ifndef ORC_SEARCH_SCHEDULE
#define ORC_SEARCH_SCHEDULE schedule(dynamic, 128)
#endif
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ threads */
int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------ KD-tree
 * [O3D] KDTreeFlann wraps nanoflann: L2, leaf_max_size 15, exact search.
 * The split rule only affects speed, never results; we use widest-dimension
 * median splits and store the points in tree order for locality. */
#define ORC_LEAF 15

typedef struct {
  int32_t left, right; /* -1 => leaf */
  int32_t begin, end;
  int32_t dim;
  double split;
} orc_node;

struct orc_kdtree {
  size_t n;
  double* pts;  /* 3n, tree order */
  int32_t* idx; /* tree position -> original index */
  orc_node* nodes;
  int32_t n_nodes, cap_nodes;
};

typedef struct {
  double p[3];
  int32_t id;
  int32_t pad;
} kd_item; /* 32 B: build works on a physically permuted copy so every pass is sequential */

static inline void swap_item(kd_item* a, kd_item* b) {
  kd_item t = *a;
  *a = *b;
  *b = t;
}

/* quickselect (Hoare partition, robust to duplicate keys): on return it[k] is the k-th smallest along dim in [lo,hi) */
static void select_kth(kd_item* it, int32_t lo, int32_t hi, int32_t k, int dim) {
  int32_t l = lo, r = hi - 1;
  while (l < r) {
    double a = it[l].p[dim], b = it[l + (r - l) / 2].p[dim], c = it[r].p[dim];
    double pivot = (a < b) ? ((b < c) ? b : ((a < c) ? c : a)) : ((a < c) ? a : ((b < c) ? c : b));
    int32_t i = l, j = r;
    while (i <= j) {
      while (it[i].p[dim] < pivot) ++i;
      while (it[j].p[dim] > pivot) --j;
      if (i <= j) {
        swap_item(&it[i], &it[j]);
        ++i;
        --j;
      }
    }
    if (k <= j)
      r = j;
    else if (k >= i)
      l = i;
    else
      return;
  }
}

static int32_t new_node(orc_kdtree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (orc_node*)realloc(t->nodes, sizeof(orc_node) * (size_t)t->cap_nodes);
  }
  return t->n_nodes++;
}

static int32_t build_rec(orc_kdtree* t, kd_item* it, int32_t lo, int32_t hi) {
  int32_t id = new_node(t);
  orc_node nd;
  nd.begin = lo;
  nd.end = hi;
  nd.left = nd.right = -1;
  nd.dim = 0;
  nd.split = 0.0;
  if (hi - lo > ORC_LEAF) {
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int32_t i = lo; i < hi; ++i) {
      const double* p = it[i].p;
      for (int d = 0; d < 3; ++d) {
        if (p[d] < mn[d]) mn[d] = p[d];
        if (p[d] > mx[d]) mx[d] = p[d];
      }
    }
    int dim = 0;
    double span = mx[0] - mn[0];
    for (int d = 1; d < 3; ++d)
      if (mx[d] - mn[d] > span) {
        span = mx[d] - mn[d];
        dim = d;
      }
    if (span > 0.0) {
      int32_t mid = lo + (hi - lo) / 2;
      select_kth(it, lo, hi, mid, dim);
      nd.dim = dim;
      nd.split = it[mid].p[dim];
      t->nodes[id] = nd;
      int32_t l = build_rec(t, it, lo, mid);
      int32_t r = build_rec(t, it, mid, hi);
      nd.left = l;
      nd.right = r;
    } /* else: all points identical -> oversized leaf */
  }
  t->nodes[id] = nd;
  return id;
}

orc_kdtree* orc_kdtree_build(const double* pts, size_t n) {
  orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof(orc_kdtree));
  t->n = n;
  t->idx = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  t->pts = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  kd_item* it = (kd_item*)malloc(sizeof(kd_item) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    memcpy(it[i].p, pts + 3 * i, 3 * sizeof(double));
    it[i].id = (int32_t)i;
  }
  if (n > 0) build_rec(t, it, 0, (int32_t)n);
  for (size_t i = 0; i < n; ++i) {
    memcpy(t->pts + 3 * i, it[i].p, 3 * sizeof(double));
    t->idx[i] = it[i].id;
  }
  free(it);
  return t;
}

void orc_kdtree_free(orc_kdtree* t) {
  if (!t) return;
  free(t->pts);
  free(t->idx);
  free(t->nodes);
  free(t);
}

typedef struct {
  int k, count;
  double worst;     /* candidates must satisfy d2 < worst (strict) ... */
  int32_t worst_id; /* ... or, once k are held, tie the k-th distance with a smaller original index */
  int32_t* idx;
  double* d2;
} knn_state;

/* Order of equal distances.  nanoflann's KNNResultSet keeps equal distances in the order the tree walk met them and lets
 * the first one met keep the k-th place, i.e. [O3D] leaves ties to the shape of its tree.  The oracle fixes what [O3D] leaves
 * open: neighbours are ordered by (d2, original index) -- a total order that does not depend on any tree (or grid), so that a
 * result can be reproduced bit for bit by an implementation with a different search structure. */
static inline int knn_accepts(const knn_state* s, double d2, int32_t id) {
  return d2 < s->worst || (s->count == s->k && d2 == s->worst && id < s->worst_id);
}

static inline void knn_push(knn_state* s, double d2, int32_t id) {
  /* insertion into a list of at most k entries, ascending by (d2, id) */
  int pos = s->count < s->k ? s->count : s->k - 1;
  while (pos > 0 && (s->d2[pos - 1] > d2 || (s->d2[pos - 1] == d2 && s->idx[pos - 1] > id))) {
    s->d2[pos] = s->d2[pos - 1];
    s->idx[pos] = s->idx[pos - 1];
    --pos;
  }
  s->d2[pos] = d2;
  s->idx[pos] = id;
  if (s->count < s->k) ++s->count;
  if (s->count == s->k) {
    s->worst = s->d2[s->k - 1];
    s->worst_id = s->idx[s->k - 1];
  }
}

static void search_rec(const orc_kdtree* t, int32_t node, const double q[3], knn_state* s) {
  const orc_node* nd = &t->nodes[node];
  if (nd->left < 0) {
    for (int32_t i = nd->begin; i < nd->end; ++i) {
      const double* p = t->pts + 3 * (size_t)i;
      double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
      double d2 = dx * dx + dy * dy + dz * dz;
      if (knn_accepts(s, d2, t->idx[i])) knn_push(s, d2, t->idx[i]);
    }
    return;
  }
  double diff = q[nd->dim] - nd->split;
  int32_t near = diff < 0 ? nd->left : nd->right;
  int32_t far = diff < 0 ? nd->right : nd->left;
  search_rec(t, near, q, s);
  if (diff * diff <= s->worst) search_rec(t, far, q, s); /* <=: a tie on the far side may carry a smaller index */
}

int orc_kdtree_search_hybrid(const orc_kdtree* t, const double q[3], double radius, int max_nn, int32_t* idx, double* d2) {
  if (!t || t->n == 0 || max_nn <= 0) return 0;
  knn_state s;
  s.k = max_nn;
  s.count = 0;
  s.worst = radius * radius; /* [O3D] kNN then keep entries with d2 < r^2 (lower_bound) */
  s.worst_id = 0;
  s.idx = idx;
  s.d2 = d2;
  search_rec(t, 0, q, &s);
  return s.count;
}

int orc_kdtree_search_knn(const orc_kdtree* t, const double q[3], int k, int32_t* idx, double* d2) {
  if (!t || t->n == 0 || k <= 0) return 0;
  knn_state s;
  s.k = k;
  s.count = 0;
  s.worst = DBL_MAX;
  s.worst_id = 0;
  s.idx = idx;
  s.d2 = d2;
  search_rec(t, 0, q, &s);
  return s.count;
}

/* ------------------------------------------------------------------ A.2
 * [O3D] GetRegistrationResultAndCorrespondences: 1-NN, accept iff d2 < r^2. */
void orc_evaluate(const orc_kdtree* t, const double* src, size_t n, double max_corr, int32_t* corr, double* d2_out, double* fitness,
                  double* inlier_rmse, uint64_t* n_corr) {
  double err2 = 0.0;
  uint64_t cnt = 0;
  if (max_corr > 0.0) {
    /* the searches are independent and of very uneven cost (a query without a neighbour within r visits the most nodes): dealt out
     * dynamically; the squared distances are summed afterwards in index order, so the result does not depend on the thread count */
    double* d2v = d2_out ? d2_out : (double*)malloc(sizeof(double) * (n ? n : 1));
#pragma omp parallel for ORC_SEARCH_SCHEDULE
    for (long i = 0; i < (long)n; ++i) {
      int32_t id;
      double d2;
      int k = orc_kdtree_search_hybrid(t, src + 3 * (size_t)i, max_corr, 1, &id, &d2);
      corr[i] = k > 0 ? id : -1;
      d2v[i] = k > 0 ? d2 : -1.0;
    }
    for (size_t i = 0; i < n; ++i)
      if (corr[i] >= 0) {
        err2 += d2v[i];
        cnt += 1;
      }
    if (!d2_out) free(d2v);
  } else {
    for (size_t i = 0; i < n; ++i) corr[i] = -1;
  }
  if (cnt == 0) {
    *fitness = 0.0;
    *inlier_rmse = 0.0;
  } else {
    *fitness = (double)cnt / (double)n;
    *inlier_rmse = sqrt(err2 / (double)cnt);
  }
  *n_corr = cnt;
}

/* ------------------------------------------------------------------ A.3
 * [O3D] TransformationEstimationPointToPlane::ComputeTransformation +
 * utility::ComputeJTJandJTr: r = (p - q).n ; J = [p x n ; n] ; w = 1 (L2 loss). */
void orc_compute_jtj_jtr(const double* src, size_t n, const double* tgt, const double* tgt_nrm, const int32_t* corr, double JTJ[36],
                         double JTr[6], double* r2_sum) {
  int nt = orc_num_threads();
  double* part = (double*)calloc((size_t)nt * 44, sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double A[36] = {0}, b[6] = {0}, r2 = 0.0;
#pragma omp for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
      int32_t j = corr[i];
      if (j < 0) continue;
      const double* p = src + 3 * (size_t)i;
      const double* q = tgt + 3 * (size_t)j;
      const double* nn = tgt_nrm + 3 * (size_t)j;
      double r = (p[0] - q[0]) * nn[0] + (p[1] - q[1]) * nn[1] + (p[2] - q[2]) * nn[2];
      double J[6] = {p[1] * nn[2] - p[2] * nn[1], p[2] * nn[0] - p[0] * nn[2], p[0] * nn[1] - p[1] * nn[0], nn[0], nn[1], nn[2]};
      for (int a = 0; a < 6; ++a) {
        for (int c = 0; c < 6; ++c) A[a * 6 + c] += J[a] * J[c];
        b[a] += J[a] * r;
      }
      r2 += r * r;
    }
    memcpy(part + (size_t)tid * 44, A, sizeof(A));
    memcpy(part + (size_t)tid * 44 + 36, b, sizeof(b));
    part[(size_t)tid * 44 + 42] = r2;
  }
  memset(JTJ, 0, 36 * sizeof(double));
  memset(JTr, 0, 6 * sizeof(double));
  double r2 = 0.0;
  for (int t = 0; t < nt; ++t) { /* fixed thread order: deterministic for a fixed thread count */
    for (int a = 0; a < 36; ++a) JTJ[a] += part[(size_t)t * 44 + a];
    for (int a = 0; a < 6; ++a) JTr[a] += part[(size_t)t * 44 + 36 + a];
    r2 += part[(size_t)t * 44 + 42];
  }
  if (r2_sum) *r2_sum = r2;
  free(part);
}

/* [O3D] TransformVector6dToMatrix4d: R = Rz(x[2]) * Ry(x[1]) * Rx(x[0]); t = x[3:6]. Column-major out. */
void orc_vector6_to_matrix4(const double x[6], double U[16]) {
  double ca = cos(x[0]), sa = sin(x[0]);
  double cb = cos(x[1]), sb = sin(x[1]);
  double cg = cos(x[2]), sg = sin(x[2]);
  double R[3][3];
  R[0][0] = cg * cb;
  R[0][1] = cg * sb * sa - sg * ca;
  R[0][2] = cg * sb * ca + sg * sa;
  R[1][0] = sg * cb;
  R[1][1] = sg * sb * sa + cg * ca;
  R[1][2] = sg * sb * ca - cg * sa;
  R[2][0] = -sb;
  R[2][1] = cb * sa;
  R[2][2] = cb * ca;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) U[c * 4 + r] = R[r][c];
  U[3] = U[7] = U[11] = 0.0;
  U[12] = x[3];
  U[13] = x[4];
  U[14] = x[5];
  U[15] = 1.0;
}

/* [O3D] SolveLinearSystemPSD(A,-b) = A.ldlt().solve(-b): LDL^T with symmetric
 * (largest-|diagonal|) pivoting as Eigen's LDLT does; no PSD/determinant check. */
int orc_solve_update(const double JTJ[36], const double JTr[6], double U[16], double x_out[6]) {
  double A[6][6], b[6], x[6];
  int perm[6];
  for (int i = 0; i < 6; ++i) {
    perm[i] = i;
    b[i] = -JTr[i];
    for (int j = 0; j < 6; ++j) A[i][j] = JTJ[i * 6 + j];
  }
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = fabs(A[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (fabs(A[i][i]) > best) {
        best = fabs(A[i][i]);
        piv = i;
      }
    if (piv != k) {
      for (int j = 0; j < 6; ++j) {
        double t = A[k][j];
        A[k][j] = A[piv][j];
        A[piv][j] = t;
      }
      for (int i = 0; i < 6; ++i) {
        double t = A[i][k];
        A[i][k] = A[i][piv];
        A[i][piv] = t;
      }
      double tb = b[k];
      b[k] = b[piv];
      b[piv] = tb;
      int tp = perm[k];
      perm[k] = perm[piv];
      perm[piv] = tp;
    }
    double d = A[k][k];
    /* Eigen ldlt_inplace: "pivot_is_valid = abs(realAkk) > 0; if (pivot_is_valid) A21 /= realAkk" -- a column under an exactly
     * zero pivot is left undivided (it is zero itself for a PSD matrix: e.g. a scene of one plane, whose in-plane translations and
     * the rotation about its normal have all-zero rows in J^T J) */
    for (int i = k + 1; i < 6; ++i) {
      double l = d != 0.0 ? A[i][k] / d : A[i][k];
      for (int j = k + 1; j < 6; ++j) A[i][j] -= l * A[k][j];
      A[i][k] = l; /* store L */
    }
  }
  /* forward: L y = b */
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int j = 0; j < i; ++j) s -= A[i][j] * y[j];
    y[i] = s;
  }
  /* D z = y ; L^T w = z */
  double w[6];
  for (int i = 5; i >= 0; --i) {
    /* Eigen LDLT::_solve_impl: "if (abs(vecD(i)) > (numeric_limits<double>::min)()) dst.row(i) /= vecD(i); else dst.row(i).setZero()":
     * the components a (near-)null pivot leaves undetermined come out as zero, not as inf / NaN */
    double s = fabs(A[i][i]) > DBL_MIN ? y[i] / A[i][i] : 0.0;
    for (int j = i + 1; j < 6; ++j) s -= A[j][i] * w[j];
    w[i] = s;
  }
  for (int i = 0; i < 6; ++i) x[perm[i]] = w[i];
  if (x_out) memcpy(x_out, x, sizeof(x));
  orc_vector6_to_matrix4(x, U);
  return 0;
}

/* ------------------------------------------------------------------ A.4 */
void orc_transform_points(double* pts, size_t n, const double T[16]) {
  for (size_t i = 0; i < n; ++i) {
    double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    double nx = T[0] * x + T[4] * y + T[8] * z + T[12];
    double ny = T[1] * x + T[5] * y + T[9] * z + T[13];
    double nz = T[2] * x + T[6] * y + T[10] * z + T[14];
    double w = T[3] * x + T[7] * y + T[11] * z + T[15];
    pts[3 * i] = nx / w;
    pts[3 * i + 1] = ny / w;
    pts[3 * i + 2] = nz / w;
  }
}
void orc_transform_normals(double* nrm, size_t n, const double T[16]) {
  for (size_t i = 0; i < n; ++i) {
    double x = nrm[3 * i], y = nrm[3 * i + 1], z = nrm[3 * i + 2];
    nrm[3 * i] = T[0] * x + T[4] * y + T[8] * z;
    nrm[3 * i + 1] = T[1] * x + T[5] * y + T[9] * z;
    nrm[3 * i + 2] = T[2] * x + T[6] * y + T[10] * z;
  }
}

static void mat4_mul(const double A[16], const double B[16], double C[16]) { /* column-major C = A*B */
  double R[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
      R[c * 4 + r] = s;
    }
  memcpy(C, R, sizeof(R));
}

static int is_identity16(const double T[16]) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      if (fabs(T[c * 4 + r] - (r == c ? 1.0 : 0.0)) > 1e-12) return 0; /* Eigen isIdentity default precision */
  return 1;
}

/* ------------------------------------------------------------------ A.1
 * [O3D] RegistrationICP as called from src/CloudRegistration.cpp:44-48. */
int orc_icp_point_to_plane(const double* src, size_t n, const double* tgt, const double* tgt_nrm, size_t N, const orc_kdtree* tree,
                           double max_corr, const double init[16], int max_iter, double rel_fitness, double rel_rmse,
                           orc_icp_result* out) {
  if (max_corr <= 0.0) return -1; /* [O3D] LogError("Invalid max_correspondence_distance.") */
  if (!tgt_nrm) return -2;        /* [O3D] point-to-plane requires target normals */
  orc_kdtree* own = NULL;
  if (!tree) { /* the reference rebuilds the KD-tree on every registerClouds call */
    own = orc_kdtree_build(tgt, N);
    tree = own;
  }
  double* P = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  memcpy(P, src, sizeof(double) * 3 * n);
  double T[16];
  memcpy(T, init, sizeof(T));
  if (!is_identity16(init)) orc_transform_points(P, n, init);
  double fit, rmse;
  uint64_t nc;
  orc_evaluate(tree, P, n, max_corr, corr, NULL, &fit, &rmse, &nc);
  int it = 0, converged = 0;
  for (int i = 0; i < max_iter; ++i) {
    double U[16], JTJ[36], JTr[6], r2;
    if (nc == 0) {
      double I6[6] = {0, 0, 0, 0, 0, 0};
      orc_vector6_to_matrix4(I6, U); /* empty correspondence set => identity update */
    } else {
      orc_compute_jtj_jtr(P, n, tgt, tgt_nrm, corr, JTJ, JTr, &r2);
      orc_solve_update(JTJ, JTr, U, NULL);
    }
    mat4_mul(U, T, T);
    orc_transform_points(P, n, U);
    double pf = fit, pr = rmse;
    orc_evaluate(tree, P, n, max_corr, corr, NULL, &fit, &rmse, &nc);
    ++it;
    if (fabs(pf - fit) < rel_fitness && fabs(pr - rmse) < rel_rmse) {
      converged = 1;
      break;
    }
  }
  memcpy(out->transformation, T, sizeof(T));
  out->fitness = fit;
  out->inlier_rmse = rmse;
  out->iterations = it;
  out->converged = converged;
  out->n_corr = nc;
  free(P);
  free(corr);
  if (own) orc_kdtree_free(own);
  return 0;
}

/* ------------------------------------------------------------------ A.3b point-to-point step
 * [O3D] TransformationEstimationPointToPoint::ComputeTransformation = Eigen::umeyama(source, target, with_scaling = false)
 * (call site src/CloudRegistration.cpp:69-74): means, Sigma = (1/m) sum (t - mu_t)(s - mu_s)^T, Sigma = U D V^T (JacobiSVD,
 * singular values descending), S = diag(1,1,-1) iff det(U) det(V) < 0, R = U S V^T, t = mu_t - R mu_s. */
static double det3(const double A[9]) { /* row-major */
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* one-sided Jacobi SVD of a 3x3 (row-major): A = U diag(d) V^T, d descending, U and V orthogonal (columns for vanishing
 * singular values are completed by cross products) */
void orc_svd3(const double A_in[9], double U[9], double d[3], double V[9]) {
  double A[9], Vm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(A, A_in, sizeof(A));
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += A[r * 3 + p] * A[r * 3 + p];
          beta += A[r * 3 + q] * A[r * 3 + q];
          gamma += A[r * 3 + p] * A[r * 3 + q];
        }
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
        off += fabs(gamma);
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < 3; ++r) {
          const double ap = A[r * 3 + p], aq = A[r * 3 + q];
          A[r * 3 + p] = c * ap - sn * aq;
          A[r * 3 + q] = sn * ap + c * aq;
          const double vp = Vm[r * 3 + p], vq = Vm[r * 3 + q];
          Vm[r * 3 + p] = c * vp - sn * vq;
          Vm[r * 3 + q] = sn * vp + c * vq;
        }
      }
    if (off == 0.0) break;
  }
  double nrm[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) nrm[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
  for (int a = 0; a < 2; ++a) /* descending */
    for (int b = a + 1; b < 3; ++b)
      if (nrm[ord[b]] > nrm[ord[a]]) {
        const int t = ord[a];
        ord[a] = ord[b];
        ord[b] = t;
      }
  const double tiny = 1e-14 * (nrm[ord[0]] > 0 ? nrm[ord[0]] : 1.0);
  for (int j = 0; j < 3; ++j) {
    const int s = ord[j];
    d[j] = nrm[s];
    for (int r = 0; r < 3; ++r) {
      V[r * 3 + j] = Vm[r * 3 + s];
      U[r * 3 + j] = nrm[s] > tiny ? A[r * 3 + s] / nrm[s] : 0.0;
    }
  }
  /* complete U for vanishing singular values (rank-deficient Sigma: collinear / coplanar-degenerate correspondences) */
  if (!(d[0] > tiny)) {
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(U, I, sizeof(I));
  } else {
    if (!(d[1] > tiny)) { /* any unit vector orthogonal to u0 */
      const double u0[3] = {U[0], U[3], U[6]};
      const int k = fabs(u0[0]) <= fabs(u0[1]) ? (fabs(u0[0]) <= fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
      double e[3] = {0, 0, 0}, w[3];
      e[k] = 1.0;
      const double dp = u0[k];
      for (int r = 0; r < 3; ++r) w[r] = e[r] - dp * u0[r];
      const double wn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      for (int r = 0; r < 3; ++r) U[r * 3 + 1] = w[r] / wn;
    }
    if (!(d[2] > tiny)) { /* u2 = u0 x u1 */
      U[2] = U[3] * U[7] - U[6] * U[4];
      U[5] = U[6] * U[1] - U[0] * U[7];
      U[8] = U[0] * U[4] - U[3] * U[1];
    }
  }
}

/* U (4x4 column-major) from the matched pairs; returns 0.  P = current (transformed) source points */
int orc_umeyama_update(const double* P, size_t n, const double* tgt, const int32_t* corr, double Uout[16]) {
  double ms[3] = {0, 0, 0}, mt[3] = {0, 0, 0};
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    if (corr[i] < 0) continue;
    const double* q = tgt + 3 * (size_t)corr[i];
    for (int k = 0; k < 3; ++k) {
      ms[k] += P[3 * i + k];
      mt[k] += q[k];
    }
    ++m;
  }
  double I6[6] = {0, 0, 0, 0, 0, 0};
  if (m == 0) { /* [O3D] corres.empty() -> Identity */
    orc_vector6_to_matrix4(I6, Uout);
    return 0;
  }
  const double inv = 1.0 / (double)m;
  for (int k = 0; k < 3; ++k) {
    ms[k] *= inv;
    mt[k] *= inv;
  }
  double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* row-major: S[a][b] = mean (t_a - mt_a)(s_b - ms_b) */
  for (size_t i = 0; i < n; ++i) {
    if (corr[i] < 0) continue;
    const double* q = tgt + 3 * (size_t)corr[i];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) S[a * 3 + b] += (q[a] - mt[a]) * (P[3 * i + b] - ms[b]);
  }
  for (int k = 0; k < 9; ++k) S[k] *= inv;
  double Um[9], d[3], Vm[9];
  orc_svd3(S, Um, d, Vm);
  const double sgn = det3(Um) * det3(Vm) < 0.0 ? -1.0 : 1.0;
  double R[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) R[a * 3 + b] = Um[a * 3] * Vm[b * 3] + Um[a * 3 + 1] * Vm[b * 3 + 1] + sgn * Um[a * 3 + 2] * Vm[b * 3 + 2];
  for (int k = 0; k < 16; ++k) Uout[k] = 0.0;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) Uout[b * 4 + a] = R[a * 3 + b]; /* column-major */
    Uout[12 + a] = mt[a] - (R[a * 3] * ms[0] + R[a * 3 + 1] * ms[1] + R[a * 3 + 2] * ms[2]);
  }
  Uout[15] = 1.0;
  return 0;
}

/* [O3D] RegistrationICP with TransformationEstimationPointToPoint (src/CloudRegistration.cpp:69-74): same driver as A.1 */
int orc_icp_point_to_point(const double* src, size_t n, const double* tgt, size_t N, const orc_kdtree* tree, double max_corr,
                           const double init[16], int max_iter, double rel_fitness, double rel_rmse, orc_icp_result* out) {
  if (max_corr <= 0.0) return -1;
  orc_kdtree* own = NULL;
  if (!tree) {
    own = orc_kdtree_build(tgt, N);
    tree = own;
  }
  double* P = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  memcpy(P, src, sizeof(double) * 3 * n);
  double T[16];
  memcpy(T, init, sizeof(T));
  if (!is_identity16(init)) orc_transform_points(P, n, init);
  double fit, rmse;
  uint64_t nc;
  orc_evaluate(tree, P, n, max_corr, corr, NULL, &fit, &rmse, &nc);
  int it = 0, converged = 0;
  for (int i = 0; i < max_iter; ++i) {
    double U[16];
    orc_umeyama_update(P, n, tgt, corr, U);
    mat4_mul(U, T, T);
    orc_transform_points(P, n, U);
    double pf = fit, pr = rmse;
    orc_evaluate(tree, P, n, max_corr, corr, NULL, &fit, &rmse, &nc);
    ++it;
    if (fabs(pf - fit) < rel_fitness && fabs(pr - rmse) < rel_rmse) {
      converged = 1;
      break;
    }
  }
  memcpy(out->transformation, T, sizeof(T));
  out->fitness = fit;
  out->inlier_rmse = rmse;
  out->iterations = it;
  out->converged = converged;
  out->n_corr = nc;
  free(P);
  free(corr);
  if (own) orc_kdtree_free(own);
  return 0;
}

/* ------------------------------------------------------------------ A.9
 * [O3D] GetInformationMatrixFromPointClouds(source, target, r, T) (call sites src/constraint_builders.cpp:70-73,
 * src/PlaceRecognition.cpp:148-149): transform the source by T, 1-NN within r as in A.2, then Lambda = sum G^T G over the matched
 * TARGET points q = (x,y,z), G = [[0,z,-y,1,0,0],[-z,0,x,0,1,0],[y,-x,0,0,0,1]].  out: row-major 6x6. */
int orc_information_matrix(const double* src, size_t n, const double* tgt, size_t N, const orc_kdtree* tree, double max_corr,
                           const double T[16], double out[36]) {
  if (max_corr <= 0.0) return -1;
  orc_kdtree* own = NULL;
  if (!tree) {
    own = orc_kdtree_build(tgt, N);
    tree = own;
  }
  double* P = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  memcpy(P, src, sizeof(double) * 3 * n);
  if (!is_identity16(T)) orc_transform_points(P, n, T);
  double fit, rmse;
  uint64_t nc;
  orc_evaluate(tree, P, n, max_corr, corr, NULL, &fit, &rmse, &nc);
  for (int k = 0; k < 36; ++k) out[k] = 0.0;
  for (size_t i = 0; i < n; ++i) {
    if (corr[i] < 0) continue;
    const double* q = tgt + 3 * (size_t)corr[i];
    const double G[3][6] = {{0, q[2], -q[1], 1, 0, 0}, {-q[2], 0, q[0], 0, 1, 0}, {q[1], -q[0], 0, 0, 0, 1}};
    for (int r = 0; r < 3; ++r)
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) out[a * 6 + b] += G[r][a] * G[r][b];
  }
  free(P);
  free(corr);
  if (own) orc_kdtree_free(own);
  return 0;
}

/* ------------------------------------------------------------------ space carving of the sparse map
 * getIdxsOfCarvedPoints (src/helpers.cpp:235-271) as called from Submap::carve (src/Submap.cpp:109-125): a VoxelMap of the map
 * points listed in `subset` (key = floor(p * (1/v)) per axis, VoxelHashMap.hpp:47-50, Voxel.cpp:123-129); every scan point
 * (already in the map frame) casts a ray from the sensor position, sampled every v from distance 0 while
 * distance < max(v, min(length - truncation, max_length)); every map point in a sampled voxel is removed if the map has no
 * normals or |direction . normalize(normal)| > min_dot.  flags_out[N] is set to 1 for removed ids; returns their number. */
typedef struct {
  int64_t key;
  int64_t idx;
} carve_item;
static int carve_cmp(const void* a, const void* b) {
  const carve_item *x = (const carve_item*)a, *y = (const carve_item*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static int64_t carve_key(const double p[3], double inv) {
  const int64_t kx = (int64_t)(int)floor(p[0] * inv), ky = (int64_t)(int)floor(p[1] * inv), kz = (int64_t)(int)floor(p[2] * inv);
  return ((kz + (1ll << 20)) << 42) | ((ky + (1ll << 20)) << 21) | (kx + (1ll << 20)); /* |k| < 2^20 per axis */
}
size_t orc_carve_flags(const double* scan, size_t n_scan, const double sensor[3], const double* map_pts, const double* map_nrm, size_t N,
                       const int64_t* subset, size_t n_subset, double voxel, double max_length, double truncation, double min_dot,
                       uint8_t* flags_out) {
  memset(flags_out, 0, N);
  if (n_subset == 0 || n_scan == 0) return 0;
  const double inv = 1.0 / voxel;
  carve_item* items = (carve_item*)malloc(sizeof(carve_item) * n_subset);
  for (size_t i = 0; i < n_subset; ++i) {
    items[i].idx = subset[i];
    items[i].key = carve_key(map_pts + 3 * (size_t)subset[i], inv);
  }
  qsort(items, n_subset, sizeof(carve_item), carve_cmp);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n_scan; ++i) {
    const double* p = scan + 3 * i;
    const double d[3] = {p[0] - sensor[0], p[1] - sensor[1], p[2] - sensor[2]};
    const double length = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(length > 0.0)) continue; /* the reference divides by zero here; such a ray samples no voxel */
    const double dir[3] = {d[0] / length, d[1] / length, d[2] / length};
    const double lim = fmax(voxel, fmin(length - truncation, max_length));
    for (double dist = 0.0; dist < lim; dist += voxel) {
      const double pos[3] = {dist * dir[0] + sensor[0], dist * dir[1] + sensor[1], dist * dir[2] + sensor[2]};
      const int64_t key = carve_key(pos, inv);
      size_t lo = 0, hi = n_subset; /* first item with key >= key */
      while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (items[mid].key < key)
          lo = mid + 1;
        else
          hi = mid;
      }
      for (size_t j = lo; j < n_subset && items[j].key == key; ++j) {
        const size_t id = (size_t)items[j].idx;
        int rem = 1;
        if (map_nrm) {
          const double* nn = map_nrm + 3 * id;
          const double nl = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
          const double dot = nl > 0.0 ? (dir[0] * nn[0] + dir[1] * nn[1] + dir[2] * nn[2]) / nl : 0.0; /* Eigen normalized(): 0 stays 0 */
          rem = fabs(dot) > min_dot;
        }
        if (rem) flags_out[id] = 1; /* idempotent, no critical section needed */
      }
    }
  }
  free(items);
  size_t cnt = 0;
  for (size_t i = 0; i < N; ++i) cnt += flags_out[i];
  return cnt;
}

/* ------------------------------------------------------------------ overlap of two clouds on a voxel grid
 * computeIndicesOfOverlappingPoints (src/helpers.cpp:307-332; call sites PlaceRecognition.cpp:103, constraint_builders.cpp:54):
 * both clouds (the source placed by T) are binned with the key floor(p * (1/v)); a voxel counts when it holds at least
 * min_points of EACH cloud, and then contributes all its source and all its target indices.  flags_*[i] = 1 for selected points
 * (the reference's output order is unordered_map iteration order; the index SETS are what is specified). */
typedef struct {
  int64_t key;
  int64_t idx; /* >= 0: target index, < 0: -(source index) - 1 */
} ovl_item;
static int ovl_cmp(const void* a, const void* b) {
  const ovl_item *x = (const ovl_item*)a, *y = (const ovl_item*)b;
  return x->key < y->key ? -1 : (x->key > y->key ? 1 : 0);
}
void orc_overlap_flags(const double* src, size_t n_src, const double* tgt, size_t n_tgt, const double T[16], double voxel, size_t min_points,
                       uint8_t* flags_src, uint8_t* flags_tgt) {
  memset(flags_src, 0, n_src);
  memset(flags_tgt, 0, n_tgt);
  const size_t n = n_src + n_tgt;
  if (n == 0) return;
  const double inv = 1.0 / voxel;
  ovl_item* it = (ovl_item*)malloc(sizeof(ovl_item) * n);
  double* P = (double*)malloc(sizeof(double) * 3 * (n_src ? n_src : 1));
  memcpy(P, src, sizeof(double) * 3 * n_src);
  orc_transform_points(P, n_src, T); /* sourceTransformed.Transform(sourceToTarget.matrix()) -- [O3D] Transform, always applied */
  for (size_t i = 0; i < n_tgt; ++i) {
    it[i].key = carve_key(tgt + 3 * i, inv);
    it[i].idx = (int64_t)i;
  }
  for (size_t i = 0; i < n_src; ++i) {
    it[n_tgt + i].key = carve_key(P + 3 * i, inv);
    it[n_tgt + i].idx = -(int64_t)i - 1;
  }
  qsort(it, n, sizeof(ovl_item), ovl_cmp);
  for (size_t b = 0; b < n;) {
    size_t e = b, cs = 0, ct = 0;
    while (e < n && it[e].key == it[b].key) {
      if (it[e].idx < 0)
        ++cs;
      else
        ++ct;
      ++e;
    }
    if (cs >= min_points && ct >= min_points)
      for (size_t j = b; j < e; ++j) {
        if (it[j].idx < 0)
          flags_src[-(it[j].idx + 1)] = 1;
        else
          flags_tgt[it[j].idx] = 1;
      }
    b = e;
  }
  free(it);
  free(P);
}

/* ------------------------------------------------------------------ dense voxel map
 * VoxelizedPointCloud (include/open3d_slam/Voxel.hpp:38-76, src/Voxel.cpp:18-114): per voxel (key floor(p * (1/v))) the running
 * sums of position and normal and the point count; toPointCloud emits sum / count per voxel.  The map after inserting a sequence
 * of clouds equals the fusion of their concatenation, so the oracle is stateless: out arrays sized >= 3n, voxels in
 * first-occurrence order (the reference's order is its unordered_map's); returns the number of voxels.  counts_out may be NULL. */
size_t orc_dense_fuse(const double* pts, const double* nrm, size_t n, double voxel, double* out_pts, double* out_nrm, int32_t* counts_out) {
  if (n == 0) return 0;
  const double inv = 1.0 / voxel;
  carve_item* it = (carve_item*)malloc(sizeof(carve_item) * n);
  for (size_t i = 0; i < n; ++i) {
    it[i].key = carve_key(pts + 3 * i, inv);
    it[i].idx = (int64_t)i;
  }
  qsort(it, n, sizeof(carve_item), carve_cmp); /* by key, then by index: sums in insertion order within a voxel */
  /* first-occurrence order of the voxels = ascending smallest member index */
  size_t m = 0;
  int64_t* first = (int64_t*)malloc(sizeof(int64_t) * n);
  size_t* start = (size_t*)malloc(sizeof(size_t) * (n + 1));
  for (size_t b = 0; b < n;) {
    size_t e = b;
    while (e < n && it[e].key == it[b].key) ++e;
    first[m] = it[b].idx;
    start[m++] = b;
    b = e;
  }
  start[m] = n;
  carve_item* ord = (carve_item*)malloc(sizeof(carve_item) * m);
  for (size_t v = 0; v < m; ++v) {
    ord[v].key = first[v];
    ord[v].idx = (int64_t)v;
  }
  qsort(ord, m, sizeof(carve_item), carve_cmp);
  for (size_t o = 0; o < m; ++o) {
    const size_t v = (size_t)ord[o].idx;
    double sp[3] = {0, 0, 0}, sn[3] = {0, 0, 0};
    const size_t cnt = start[v + 1] - start[v];
    for (size_t j = start[v]; j < start[v + 1]; ++j) {
      const size_t i = (size_t)it[j].idx;
      for (int k = 0; k < 3; ++k) {
        sp[k] += pts[3 * i + k];
        if (nrm) sn[k] += nrm[3 * i + k];
      }
    }
    for (int k = 0; k < 3; ++k) {
      out_pts[3 * o + k] = sp[k] / (double)cnt; /* AggregatedVoxel::getAggregatedPosition, Voxel.cpp:18-20 */
      if (nrm && out_nrm) out_nrm[3 * o + k] = sn[k] / (double)cnt; /* getAggregatedNormal: NOT re-normalised, Voxel.cpp:21-23 */
    }
    if (counts_out) counts_out[o] = (int32_t)cnt;
  }
  free(it);
  free(first);
  free(start);
  free(ord);
  return m;
}

/* ------------------------------------------------------------------ constant-velocity de-skew
 * ConstantVelocityMotionCompensation::undistortInputPointCloud (src/MotionCompensation.cpp:64-118) with computePhase (:120-139):
 * phase = azimuth / 2 pi in [0, 1] (1 - that for a clockwise-spinning sensor; 0 for azimuth exactly 0), every point is moved by
 * the motion accumulated over phase * scan_duration at constant velocity: p' = T(phase * D * v, Rz Ry Rx(phase * D * w)) p.
 * The velocities come from the pose buffer on the host (estimateLinearAndAngularVelocity, :33-58) and are inputs here. */
void orc_undistort(double* pts, size_t n, const double lin_vel[3], const double ang_vel_rpy[3], double scan_duration, int clockwise) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) {
    double* p = pts + 3 * i;
    const double angle = atan2(p[1], p[0]);
    const double kPi = 3.14159265358979323846; /* M_PI: not in strict C11 */
    const double wrapped = angle < 0.0 ? angle + 2.0 * kPi : angle;
    double phase = 0.0;
    if (wrapped != 0.0) phase = clockwise ? 1.0 - wrapped / (2.0 * kPi) : wrapped / (2.0 * kPi);
    const double s = phase * scan_duration;
    const double x6[6] = {s * ang_vel_rpy[0], s * ang_vel_rpy[1], s * ang_vel_rpy[2], s * lin_vel[0], s * lin_vel[1], s * lin_vel[2]};
    double U[16];
    orc_vector6_to_matrix4(x6, U); /* R = Rz(yaw) Ry(pitch) Rx(roll) = fromRPY (math.cpp:32-37), t = xyz */
    const double x = p[0], y = p[1], z = p[2];
    p[0] = U[0] * x + U[4] * y + U[8] * z + U[12];
    p[1] = U[1] * x + U[5] * y + U[9] * z + U[13];
    p[2] = U[2] * x + U[6] * y + U[10] * z + U[14];
  }
}

/* ------------------------------------------------------------------ carving of the dense voxel map
 * Submap::carve for the VoxelizedPointCloud (src/Submap.cpp:126-136) = removeDuplicatePointsWithinSameVoxels (src/Voxel.cpp:162-191,
 * first point of every voxel, key floor(p * (1/v))) + getKeysOfCarvedPoints (src/helpers.cpp:347-377): every kept scan point casts
 * a ray from the sensor position, sampled every 2 * radius while distance < max(2 * radius, min(length - truncation, max_length));
 * at every sample the voxels of getVoxelsWithinPointNeighborhood (src/VoxelHashMap.cpp:13-44: test points on a lattice of pitch v
 * in [-radius, radius]^3 around the sample, kept when they lie within radius of the centre of their OWN voxel -- keys by
 * floor(p / v), the dividing variant --, plus the sample's own voxel) are removed from the map if present.
 * map_keys: packed keys (carve_key layout) of the occupied voxels, any order; removed_out[n_keys] = 1 for removed voxels. */
static int64_t carve_key_div(const double p[3], double v) {
  const int64_t kx = (int64_t)(int)floor(p[0] / v), ky = (int64_t)(int)floor(p[1] / v), kz = (int64_t)(int)floor(p[2] / v);
  return ((kz + (1ll << 20)) << 42) | ((ky + (1ll << 20)) << 21) | (kx + (1ll << 20));
}
int64_t orc_voxel_key(const double p[3], double voxel) { return carve_key(p, 1.0 / voxel); }
/* mk: map keys sorted ascending with their original positions; marks the voxel `key` as removed if the map has it (idempotent) */
static void dense_mark_removed(const carve_item* mk, size_t n_keys, int64_t key, uint8_t* removed_out) {
  size_t lo = 0, hi = n_keys;
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (mk[mid].key < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < n_keys && mk[lo].key == key) removed_out[mk[lo].idx] = 1;
}

size_t orc_dense_carve(const double* scan, size_t n_scan, const double sensor[3], const int64_t* map_keys, size_t n_keys, double voxel,
                       double radius, double max_length, double truncation, uint8_t* removed_out) {
  memset(removed_out, 0, n_keys);
  /* radius <= 0 makes the ray step 2 * radius zero: the reference's own while loop (helpers.cpp:356-372) then never terminates.
   * Not a usable input there, rejected here (SIZE_MAX). */
  if (!(radius > 0.0) || !(voxel > 0.0)) return (size_t)-1;
  if (n_keys == 0 || n_scan == 0) return 0;
  const double inv = 1.0 / voxel;
  /* sorted copy of the map keys for the lookups */
  carve_item* mk = (carve_item*)malloc(sizeof(carve_item) * n_keys);
  for (size_t i = 0; i < n_keys; ++i) {
    mk[i].key = map_keys[i];
    mk[i].idx = (int64_t)i;
  }
  qsort(mk, n_keys, sizeof(carve_item), carve_cmp);
  /* removeDuplicatePointsWithinSameVoxels: first occurrence per voxel */
  carve_item* sk = (carve_item*)malloc(sizeof(carve_item) * n_scan);
  for (size_t i = 0; i < n_scan; ++i) {
    sk[i].key = carve_key(scan + 3 * i, inv);
    sk[i].idx = (int64_t)i;
  }
  qsort(sk, n_scan, sizeof(carve_item), carve_cmp); /* by key then index */
  uint8_t* keep = (uint8_t*)calloc(n_scan, 1);
  for (size_t i = 0; i < n_scan; ++i)
    if (i == 0 || sk[i].key != sk[i - 1].key) keep[sk[i].idx] = 1;
  const double step = 2.0 * radius;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n_scan; ++i) {
    if (!keep[i]) continue;
    const double* p = scan + 3 * i;
    const double d[3] = {p[0] - sensor[0], p[1] - sensor[1], p[2] - sensor[2]};
    const double length = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(length > 0.0)) continue; /* the reference divides by zero here */
    const double dir[3] = {d[0] / length, d[1] / length, d[2] / length};
    const double lim = fmax(step, fmin(length - truncation, max_length));
    for (double dist = 0.0; dist < lim; dist += step) {
      const double pos[3] = {dist * dir[0] + sensor[0], dist * dir[1] + sensor[1], dist * dir[2] + sensor[2]};
      const int64_t center_key = carve_key_div(pos, voxel);
      int center_added = 0; /* (the radius <= 0 branch of getVoxelsWithinPointNeighborhood is unreachable from here, see above) */
      for (double dx = -radius; dx <= radius; dx += voxel)
        for (double dy = -radius; dy <= radius; dy += voxel)
          for (double dz = -radius; dz <= radius; dz += voxel) {
            const double tp[3] = {pos[0] + dx, pos[1] + dy, pos[2] + dz};
            /* getCenterOfCorrespondingVoxel: key (by division) * v + v / 2 */
            const double c[3] = {floor(tp[0] / voxel) * voxel + voxel * 0.5, floor(tp[1] / voxel) * voxel + voxel * 0.5,
                                 floor(tp[2] / voxel) * voxel + voxel * 0.5};
            const double e[3] = {tp[0] - c[0], tp[1] - c[1], tp[2] - c[2]};
            if (sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) <= radius) {
              const int64_t key = carve_key_div(tp, voxel);
              if (key == center_key) center_added = 1;
              dense_mark_removed(mk, n_keys, key, removed_out);
            }
          }
      if (!center_added) dense_mark_removed(mk, n_keys, center_key, removed_out); /* VoxelHashMap.cpp:40-42 */
    }
  }
  free(mk);
  free(sk);
  free(keep);
  size_t cnt = 0;
  for (size_t i = 0; i < n_keys; ++i) cnt += removed_out[i];
  return cnt;
}

/* ------------------------------------------------------------------ A.8 Generalized ICP
 * [O3D] GeneralizedICP.cpp: GetRotationFromE1ToX, InitializePointCloudForGeneralizedICP,
 * TransformationEstimationForGeneralizedICP::ComputeTransformation; reference call site src/CloudRegistration.cpp:16-21. */
static void mat3_mul(const double A[9], const double B[9], double C[9]) { /* row-major */
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, R, sizeof(R));
}
static void mat3_t(const double A[9], double T[9]) {
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = A[j * 3 + i];
  memcpy(T, R, sizeof(R));
}

void orc_covariance_from_normal(const double x[3], double epsilon, double cov[9]) {
  /* Rx = GetRotationFromE1ToX(x): v = e1 x x, c = e1 . x; c < -0.99 -> Identity; else I + [v]x + [v]x^2 / (1 + c) */
  double Rx[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double c = x[0];
  if (!(c < -0.99)) {
    const double v[3] = {0.0, -x[2], x[1]}; /* e1 x x */
    const double sv[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    double sv2[9];
    mat3_mul(sv, sv, sv2);
    const double f = 1.0 / (1.0 + c);
    for (int i = 0; i < 9; ++i) Rx[i] += sv[i] + sv2[i] * f;
  }
  const double C[9] = {epsilon, 0, 0, 0, 1, 0, 0, 0, 1};
  double RxT[9], t[9];
  mat3_t(Rx, RxT);
  mat3_mul(Rx, C, t);
  mat3_mul(t, RxT, cov);
}

/* symmetric 3x3: W = M^-1/2 via Jacobi eigen-decomposition (Eigen: M.inverse().sqrt()) */
static void sym3_eig(const double Min[9], double w[3], double V[9]) {
  double A[9];
  memcpy(A, Min, sizeof(A));
  double Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[p * 3 + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* A <- A * G */
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- G^T * A */
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double qkp = Q[k * 3 + p], qkq = Q[k * 3 + q];
          Q[k * 3 + p] = c * qkp - s * qkq;
          Q[k * 3 + q] = s * qkp + c * qkq;
        }
      }
  }
  w[0] = A[0];
  w[1] = A[4];
  w[2] = A[8];
  memcpy(V, Q, sizeof(Q));
}
static void sym3_inv_sqrt(const double M[9], double W[9]) {
  double w[3], V[9];
  sym3_eig(M, w, V);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += V[i * 3 + k] * (1.0 / sqrt(w[k])) * V[j * 3 + k];
      W[i * 3 + j] = s;
    }
}

void orc_gicp_jtj_jtr(const double* src, const double* src_cov, size_t n, const double* tgt, const double* tgt_cov, const int32_t* corr,
                      double JTJ[36], double JTr[6]) {
  int nt = orc_num_threads();
  double* part = (double*)calloc((size_t)nt * 44, sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double A[36] = {0}, b[6] = {0};
#pragma omp for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
      int32_t j = corr[i];
      if (j < 0) continue;
      const double* vs = src + 3 * (size_t)i;
      const double* vt = tgt + 3 * (size_t)j;
      const double d[3] = {vs[0] - vt[0], vs[1] - vt[1], vs[2] - vt[2]};
      double M[9], W[9];
      for (int k = 0; k < 9; ++k) M[k] = tgt_cov[9 * (size_t)j + k] + src_cov[9 * (size_t)i + k];
      sym3_inv_sqrt(M, W);
      /* J = W * [ -[vs]x | I ] (3x6) */
      const double S[9] = {0, vs[2], -vs[1], -vs[2], 0, vs[0], vs[1], -vs[0], 0}; /* -SkewMatrix(vs) */
      double J[3][6];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          J[r][c] = W[r * 3] * S[c] + W[r * 3 + 1] * S[3 + c] + W[r * 3 + 2] * S[6 + c];
          J[r][3 + c] = W[r * 3 + c];
        }
      for (int r = 0; r < 3; ++r) {
        const double res = W[r * 3] * d[0] + W[r * 3 + 1] * d[1] + W[r * 3 + 2] * d[2];
        for (int a = 0; a < 6; ++a) {
          for (int c = 0; c < 6; ++c) A[a * 6 + c] += J[r][a] * J[r][c];
          b[a] += J[r][a] * res;
        }
      }
    }
    memcpy(part + (size_t)tid * 44, A, sizeof(A));
    memcpy(part + (size_t)tid * 44 + 36, b, sizeof(b));
  }
  memset(JTJ, 0, 36 * sizeof(double));
  memset(JTr, 0, 6 * sizeof(double));
  for (int t = 0; t < nt; ++t) {
    for (int a = 0; a < 36; ++a) JTJ[a] += part[(size_t)t * 44 + a];
    for (int a = 0; a < 6; ++a) JTr[a] += part[(size_t)t * 44 + 36 + a];
  }
  free(part);
}

static void transform_covs(double* cov, size_t n, const double T[16]) { /* [O3D] PointCloud::Transform: R C R^T */
  const double R[9] = {T[0], T[4], T[8], T[1], T[5], T[9], T[2], T[6], T[10]};
  double RT[9];
  mat3_t(R, RT);
  for (size_t i = 0; i < n; ++i) {
    double t[9];
    mat3_mul(R, cov + 9 * i, t);
    mat3_mul(t, RT, cov + 9 * i);
  }
}

int orc_icp_generalized(const double* src, const double* src_nrm, size_t n, const double* tgt, const double* tgt_nrm, size_t N,
                        const orc_kdtree* tree, double max_corr, const double init[16], int max_iter, double rel_fitness, double rel_rmse,
                        double epsilon, orc_icp_result* out) {
  if (max_corr <= 0.0) return -1;
  if (!src_nrm || !tgt_nrm) return -2; /* the reference always reaches this call with normals on both clouds */
  orc_kdtree* own = NULL;
  if (!tree) {
    own = orc_kdtree_build(tgt, N);
    tree = own;
  }
  double* P = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  double* Cs = (double*)malloc(sizeof(double) * 9 * (n ? n : 1));
  double* Ct = (double*)malloc(sizeof(double) * 9 * (N ? N : 1));
  int32_t* corr = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  memcpy(P, src, sizeof(double) * 3 * n);
  for (size_t i = 0; i < n; ++i) orc_covariance_from_normal(src_nrm + 3 * i, epsilon, Cs + 9 * i);
  for (size_t i = 0; i < N; ++i) orc_covariance_from_normal(tgt_nrm + 3 * i, epsilon, Ct + 9 * i);
  double T[16];
  memcpy(T, init, sizeof(T));
  if (!is_identity16(init)) {
    orc_transform_points(P, n, init);
    transform_covs(Cs, n, init);
  }
  double fit, rmse;
  uint64_t nc;
  orc_evaluate(tree, P, n, max_corr, corr, NULL, &fit, &rmse, &nc);
  int it = 0, converged = 0;
  for (int i = 0; i < max_iter; ++i) {
    double U[16], JTJ[36], JTr[6];
    if (nc == 0) {
      double z[6] = {0, 0, 0, 0, 0, 0};
      orc_vector6_to_matrix4(z, U);
    } else {
      orc_gicp_jtj_jtr(P, Cs, n, tgt, Ct, corr, JTJ, JTr);
      orc_solve_update(JTJ, JTr, U, NULL);
    }
    mat4_mul(U, T, T);
    orc_transform_points(P, n, U);
    transform_covs(Cs, n, U);
    double pf = fit, pr = rmse;
    orc_evaluate(tree, P, n, max_corr, corr, NULL, &fit, &rmse, &nc);
    ++it;
    if (fabs(pf - fit) < rel_fitness && fabs(pr - rmse) < rel_rmse) {
      converged = 1;
      break;
    }
  }
  memcpy(out->transformation, T, sizeof(T));
  out->fitness = fit;
  out->inlier_rmse = rmse;
  out->iterations = it;
  out->converged = converged;
  out->n_corr = nc;
  free(P);
  free(Cs);
  free(Ct);
  free(corr);
  if (own) orc_kdtree_free(own);
  return 0;
}

/* ------------------------------------------------------------------ A.5
 * [O3D] FastEigen3x3 (Geometric Tools "robust eigensolver for 3x3 symmetric
 * matrices"): returns the eigenvector of the smallest eigenvalue. */
/* std::acos / std::cos of FastEigen3x3, made portable.  [O3D] calls the platform's libm, whose last bit differs between
 * platforms (and from a GPU's math library); where two eigenvalues of a neighbourhood nearly coincide that bit decides which
 * eigenvector comes out.  The oracle therefore spells the two functions out with +, -, *, / and sqrt only (IEEE-754: the same
 * bits on every conforming machine when compiled without contraction, -ffp-contract=off): Taylor / binomial series in Horner
 * form on reduced arguments, coefficients = exact rationals rounded to double (scripts/gen_det_trig.py).  Within 2 ulp of libm
 * (tests/test_oracle.py::test_portable_acos_cos_agree_with_libm).  The device code evaluates the same expressions in the same
 * order (open3d_slam_amd/csrc/det_math.hpp). */
static const double kAsinC[27] = {
    0x1.5555555555555p-3, 0x1.3333333333333p-4, 0x1.6db6db6db6db7p-5, 0x1.f1c71c71c71c7p-6, 0x1.6e8ba2e8ba2e9p-6, 0x1.1c4ec4ec4ec4fp-6,
    0x1.c99999999999ap-7, 0x1.7a87878787878p-7, 0x1.3fde50d79435ep-7, 0x1.12ef3cf3cf3cfp-7, 0x1.df3bd37a6f4dfp-8, 0x1.a6863d70a3d71p-8,
    0x1.782dda12f684cp-8, 0x1.51ba308d3dcb1p-8, 0x1.31683bdef7bdfp-8, 0x1.15ee9d45d1746p-8, 0x1.fcaf8fb6db6dbp-9, 0x1.d3d2a8e0dd67dp-9,
    0x1.b026f57b13b14p-9, 0x1.90cb77f60c7cep-9, 0x1.750de64d7d05fp-9, 0x1.5c5f56efaaaabp-9, 0x1.464c0950f7d47p-9, 0x1.3275586c5f2f0p-9,
    0x1.208d3570ae5a6p-9, 0x1.1052bc5fa960ap-9, 0x1.018f963c229bfp-9};
static const double kCosC[11] = {-0x1.0000000000000p-1, 0x1.5555555555555p-5,  -0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-16,
                                 -0x1.27e4fb7789f5cp-22, 0x1.1eed8eff8d898p-29, -0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-45,
                                 -0x1.6827863b97d97p-53, 0x1.e542ba4020225p-62, -0x1.0ce396db7f853p-70};
static const double kSinC[10] = {-0x1.5555555555555p-3,  0x1.1111111111111p-7,  -0x1.a01a01a01a01ap-13, 0x1.71de3a556c734p-19,
                                 -0x1.ae64567f544e4p-26, 0x1.6124613a86d09p-33, -0x1.ae7f3e733b81fp-41, 0x1.952c77030ad4ap-49,
                                 -0x1.2f49b46814157p-57, 0x1.71b8ef6dcf572p-66};
#define ORC_PIO2_HI 0x1.921fb54442d18p+0
#define ORC_PIO2_LO 0x1.1a62633145c07p-54
#define ORC_PI_HI 0x1.921fb54442d18p+1
#define ORC_PI_LO 0x1.1a62633145c07p-53
#define ORC_PIO4 0x1.921fb54442d18p-1
#define ORC_PI3O4 0x1.2d97c7f3321d2p+1

static double asin_tail(double z) { /* P(z): asin(s) = s + s*z*P(z), z = s*s <= 1/4 */
  double p = kAsinC[26];
  for (int k = 25; k >= 0; --k) p = p * z + kAsinC[k];
  return p;
}
double orc_acos(double x) { /* x in [-1, 1] */
  if (x >= 0.5) {
    double z = (1.0 - x) * 0.5, s = sqrt(z);
    return 2.0 * (s + s * z * asin_tail(z));
  }
  if (x <= -0.5) {
    double z = (1.0 + x) * 0.5, s = sqrt(z);
    return (ORC_PI_HI - 2.0 * (s + s * z * asin_tail(z))) + ORC_PI_LO;
  }
  double z = x * x;
  return (ORC_PIO2_HI - (x + x * z * asin_tail(z))) + ORC_PIO2_LO;
}
static double cos_series(double y) { /* |y| <= pi/4 */
  double w = y * y, p = kCosC[10];
  for (int k = 9; k >= 0; --k) p = p * w + kCosC[k];
  return 1.0 + w * p;
}
static double sin_series(double y) { /* |y| <= pi/4 */
  double w = y * y, p = kSinC[9];
  for (int k = 8; k >= 0; --k) p = p * w + kSinC[k];
  return y + y * w * p;
}
double orc_cos(double x) { /* x in [0, pi] */
  if (x <= ORC_PIO4) return cos_series(x);
  if (x < ORC_PI3O4) return sin_series((ORC_PIO2_HI - x) + ORC_PIO2_LO);
  return -cos_series((ORC_PI_HI - x) + ORC_PI_LO);
}

static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void eigvec0(const double A[9], double ev, double out[3]) {
  double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[4] - ev, A[5]}, r2[3] = {A[2], A[5], A[8] - ev};
  double c01[3], c02[3], c12[3];
  cross3(r0, r1, c01);
  cross3(r0, r2, c02);
  cross3(r1, r2, c12);
  double d0 = dot3(c01, c01), d1 = dot3(c02, c02), d2 = dot3(c12, c12);
  const double* best = c01;
  double dm = d0;
  if (d1 > dm) {
    dm = d1;
    best = c02;
  }
  if (d2 > dm) {
    dm = d2;
    best = c12;
  }
  double s = sqrt(dm);
  out[0] = best[0] / s;
  out[1] = best[1] / s;
  out[2] = best[2] / s;
}

static void eigvec1(const double A[9], const double e0[3], double ev, double out[3]) {
  double U[3], V[3];
  if (fabs(e0[0]) > fabs(e0[1])) {
    double inv = 1.0 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]);
    U[0] = -e0[2] * inv;
    U[1] = 0.0;
    U[2] = e0[0] * inv;
  } else {
    double inv = 1.0 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]);
    U[0] = 0.0;
    U[1] = e0[2] * inv;
    U[2] = -e0[1] * inv;
  }
  cross3(e0, U, V);
  double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[4] * U[1] + A[5] * U[2], A[2] * U[0] + A[5] * U[1] + A[8] * U[2]};
  double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[4] * V[1] + A[5] * V[2], A[2] * V[0] + A[5] * V[1] + A[8] * V[2]};
  double m00 = dot3(U, AU) - ev, m01 = dot3(U, AV), m11 = dot3(V, AV) - ev;
  double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
  if (a00 >= a11) {
    double mx = a00 > a01 ? a00 : a01;
    if (mx > 0) {
      if (a00 >= a01) {
        m01 /= m00;
        m00 = 1.0 / sqrt(1 + m01 * m01);
        m01 *= m00;
      } else {
        m00 /= m01;
        m01 = 1.0 / sqrt(1 + m00 * m00);
        m00 *= m01;
      }
      for (int i = 0; i < 3; ++i) out[i] = m01 * U[i] - m00 * V[i];
    } else {
      for (int i = 0; i < 3; ++i) out[i] = U[i];
    }
  } else {
    double mx = a11 > a01 ? a11 : a01;
    if (mx > 0) {
      if (a11 >= a01) {
        m01 /= m11;
        m11 = 1.0 / sqrt(1 + m01 * m01);
        m01 *= m11;
      } else {
        m11 /= m01;
        m01 = 1.0 / sqrt(1 + m11 * m11);
        m11 *= m01;
      }
      for (int i = 0; i < 3; ++i) out[i] = m11 * U[i] - m01 * V[i];
    } else {
      for (int i = 0; i < 3; ++i) out[i] = U[i];
    }
  }
}

void orc_fast_eigen3x3_min_evec(const double cov[9], double out[3]) {
  double A[9];
  double mc = cov[0];
  for (int i = 1; i < 9; ++i)
    if (cov[i] > mc) mc = cov[i];
  if (mc == 0.0) {
    out[0] = out[1] = out[2] = 0.0;
    return;
  }
  for (int i = 0; i < 9; ++i) A[i] = cov[i] / mc;
  double norm = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
  if (norm > 0.0) {
    double q = (A[0] + A[4] + A[8]) / 3.0;
    double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
    double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
    double c00 = b11 * b22 - A[5] * A[5];
    double c01 = A[1] * b22 - A[5] * A[2];
    double c02 = A[1] * A[5] - b11 * A[2];
    double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
    double half_det = det * 0.5;
    if (half_det < -1.0) half_det = -1.0;
    if (half_det > 1.0) half_det = 1.0;
    double angle = orc_acos(half_det) / 3.0;
    const double two_thirds_pi = 2.09439510239319549;
    double beta2 = orc_cos(angle) * 2.0;
    double beta0 = orc_cos(angle + two_thirds_pi) * 2.0;
    double beta1 = -(beta0 + beta2);
    double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
    double v0[3], v1[3], v2[3];
    if (half_det >= 0.0) {
      eigvec0(A, e2, v2);
      if (e2 < e0 && e2 < e1) {
        memcpy(out, v2, sizeof(v2));
        return;
      }
      eigvec1(A, v2, e1, v1);
      if (e1 < e0 && e1 < e2) {
        memcpy(out, v1, sizeof(v1));
        return;
      }
      cross3(v1, v2, v0);
      memcpy(out, v0, sizeof(v0));
    } else {
      eigvec0(A, e0, v0);
      if (e0 < e1 && e0 < e2) {
        memcpy(out, v0, sizeof(v0));
        return;
      }
      eigvec1(A, v0, e1, v1);
      if (e1 < e0 && e1 < e2) {
        memcpy(out, v1, sizeof(v1));
        return;
      }
      cross3(v0, v1, v2);
      memcpy(out, v2, sizeof(v2));
    }
  } else { /* diagonal */
    if (cov[0] < cov[4] && cov[0] < cov[8]) {
      out[0] = 1;
      out[1] = 0;
      out[2] = 0;
    } else if (cov[4] < cov[0] && cov[4] < cov[8]) {
      out[0] = 0;
      out[1] = 1;
      out[2] = 0;
    } else {
      out[0] = 0;
      out[1] = 0;
      out[2] = 1;
    }
  }
}

/* [O3D] EstimatePerPointCovariances + ComputeNormal + NormalizeNormals +
 * OrientNormalsTowardsCameraLocation(0,0,0); call site src/CloudRegistration.cpp:49-56 */
void orc_estimate_normals_ex(const double* pts, size_t n, double radius, int max_nn, int raw, double* normals);
void orc_estimate_normals(const double* pts, size_t n, double radius, int max_nn, double* normals) {
  orc_estimate_normals_ex(pts, n, radius, max_nn, 0, normals);
}

/* radius <= 0: KDTreeSearchParamKNN(max_nn) instead of the hybrid search.  raw != 0: [O3D] EstimateNormals alone (ComputeNormal's vector,
 * (0,0,1) when it is zero) without NormalizeNormals / OrientNormalsTowardsCameraLocation -- what
 * InitializePointCloudForGeneralizedICP does to a cloud that has no normals (KNN(20)), call site src/CloudRegistration.cpp:16-21 */
void orc_estimate_normals_ex(const double* pts, size_t n, double radius, int max_nn, int raw, double* normals) {
  orc_kdtree* t = orc_kdtree_build(pts, n);
#pragma omp parallel
  {
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(max_nn > 0 ? max_nn : 1));
    double* d2 = (double*)malloc(sizeof(double) * (size_t)(max_nn > 0 ? max_nn : 1));
#pragma omp for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
      const double* p = pts + 3 * (size_t)i;
      int k = radius > 0.0 ? orc_kdtree_search_hybrid(t, p, radius, max_nn, idx, d2) : orc_kdtree_search_knn(t, p, max_nn, idx, d2);
      double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      if (k >= 3) {
        double c[9] = {0};
        for (int j = 0; j < k; ++j) {
          const double* v = pts + 3 * (size_t)idx[j];
          c[0] += v[0];
          c[1] += v[1];
          c[2] += v[2];
          c[3] += v[0] * v[0];
          c[4] += v[0] * v[1];
          c[5] += v[0] * v[2];
          c[6] += v[1] * v[1];
          c[7] += v[1] * v[2];
          c[8] += v[2] * v[2];
        }
        for (int j = 0; j < 9; ++j) c[j] /= (double)k;
        cov[0] = c[3] - c[0] * c[0];
        cov[4] = c[6] - c[1] * c[1];
        cov[8] = c[8] - c[2] * c[2];
        cov[1] = cov[3] = c[4] - c[0] * c[1];
        cov[2] = cov[6] = c[5] - c[0] * c[2];
        cov[5] = cov[7] = c[7] - c[1] * c[2];
      }
      double nv[3];
      orc_fast_eigen3x3_min_evec(cov, nv);
      double nn = sqrt(dot3(nv, nv));
      if (nn == 0.0) {
        nv[0] = 0;
        nv[1] = 0;
        nv[2] = 1;
        nn = 1.0;
      }
      if (raw) {
        normals[3 * (size_t)i] = nv[0];
        normals[3 * (size_t)i + 1] = nv[1];
        normals[3 * (size_t)i + 2] = nv[2];
        continue;
      }
      /* NormalizeNormals */
      nv[0] /= nn;
      nv[1] /= nn;
      nv[2] /= nn;
      if (isnan(nv[0])) {
        nv[0] = 0;
        nv[1] = 0;
        nv[2] = 1;
      }
      /* OrientNormalsTowardsCameraLocation(camera = 0): reference = -p */
      double ref[3] = {-p[0], -p[1], -p[2]};
      if (dot3(nv, nv) == 0.0) {
        double rn = sqrt(dot3(ref, ref));
        if (rn == 0.0) {
          nv[0] = 0;
          nv[1] = 0;
          nv[2] = 1;
        } else {
          nv[0] = ref[0] / rn;
          nv[1] = ref[1] / rn;
          nv[2] = ref[2] / rn;
        }
      } else if (dot3(nv, ref) < 0.0) {
        nv[0] = -nv[0];
        nv[1] = -nv[1];
        nv[2] = -nv[2];
      }
      normals[3 * (size_t)i] = nv[0];
      normals[3 * (size_t)i + 1] = nv[1];
      normals[3 * (size_t)i + 2] = nv[2];
    }
    free(idx);
    free(d2);
  }
  orc_kdtree_free(t);
}

/* ------------------------------------------------------------------ voxel hash (oracle-private) */
typedef struct {
  size_t cap; /* power of two */
  int32_t* keys; /* 3*cap */
  int64_t* slot; /* cap, -1 empty */
} vhash;

static void vhash_init(vhash* h, size_t n) {
  size_t cap = 16;
  while (cap < 2 * n + 1) cap <<= 1;
  h->cap = cap;
  h->keys = (int32_t*)malloc(sizeof(int32_t) * 3 * cap);
  h->slot = (int64_t*)malloc(sizeof(int64_t) * cap);
  for (size_t i = 0; i < cap; ++i) h->slot[i] = -1;
}
static void vhash_free(vhash* h) {
  free(h->keys);
  free(h->slot);
}
/* returns slot id; *is_new set when inserted with next_id */
static int64_t vhash_get(vhash* h, int32_t x, int32_t y, int32_t z, int64_t next_id, int* is_new) {
  /* probe start: the reference's own hash, include/open3d_slam/VoxelHashMap.hpp:25-35 */
  uint32_t hv = (uint32_t)((int64_t)x + (int64_t)y * 17191LL + (int64_t)z * 17191LL * 17191LL);
  size_t pos = (size_t)(hv * 2654435761u) & (h->cap - 1);
  for (;;) {
    if (h->slot[pos] < 0) {
      h->slot[pos] = next_id;
      h->keys[3 * pos] = x;
      h->keys[3 * pos + 1] = y;
      h->keys[3 * pos + 2] = z;
      *is_new = 1;
      return next_id;
    }
    if (h->keys[3 * pos] == x && h->keys[3 * pos + 1] == y && h->keys[3 * pos + 2] == z) {
      *is_new = 0;
      return h->slot[pos];
    }
    pos = (pos + 1) & (h->cap - 1);
  }
}

/* [O3D] PointCloud::VoxelDownSample via src/helpers.cpp:107-113 */
size_t orc_voxel_down_sample(const double* pts, const double* nrm, size_t n, double voxel, double* out_pts, double* out_nrm) {
  if (n == 0) return 0;
  if (voxel <= 0.0) { /* wrapper returns the cloud unchanged, helpers.cpp:108-110 */
    memcpy(out_pts, pts, sizeof(double) * 3 * n);
    if (nrm && out_nrm) memcpy(out_nrm, nrm, sizeof(double) * 3 * n);
    return n;
  }
  double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX};
  for (size_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d)
      if (pts[3 * i + d] < mn[d]) mn[d] = pts[3 * i + d];
  for (int d = 0; d < 3; ++d) mn[d] -= voxel * 0.5; /* voxel_min_bound = min_bound - voxel/2 */
  vhash h;
  vhash_init(&h, n);
  int64_t* cnt = (int64_t*)calloc(n, sizeof(int64_t));
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) {
    int32_t k[3];
    for (int d = 0; d < 3; ++d) k[d] = (int32_t)floor((pts[3 * i + d] - mn[d]) / voxel);
    int is_new;
    int64_t s = vhash_get(&h, k[0], k[1], k[2], (int64_t)m, &is_new);
    if (is_new) {
      for (int d = 0; d < 3; ++d) {
        out_pts[3 * m + d] = 0.0;
        if (nrm && out_nrm) out_nrm[3 * m + d] = 0.0;
      }
      ++m;
    }
    for (int d = 0; d < 3; ++d) {
      out_pts[3 * (size_t)s + d] += pts[3 * i + d];
      if (nrm && out_nrm) out_nrm[3 * (size_t)s + d] += nrm[3 * i + d];
    }
    cnt[s] += 1;
  }
  for (size_t s = 0; s < m; ++s)
    for (int d = 0; d < 3; ++d) {
      out_pts[3 * s + d] /= (double)cnt[s];
      if (nrm && out_nrm) out_nrm[3 * s + d] /= (double)cnt[s]; /* [O3D] normals averaged, not re-normalised */
    }
  free(cnt);
  vhash_free(&h);
  return m;
}

/* ------------------------------------------------------------------ croppers: src/croppers.cpp:121-165 */
static int within_volume(const double* p, const orc_crop* c) {
  int in = 1;
  double dx = p[0] - c->center[0], dy = p[1] - c->center[1], dz = p[2] - c->center[2];
  switch (c->kind) {
    case ORC_CROP_MAX_RADIUS:
      in = sqrt(dx * dx + dy * dy + dz * dz) <= c->rmax; /* croppers.cpp:136-138 */
      break;
    case ORC_CROP_MIN_RADIUS:
      in = sqrt(dx * dx + dy * dy + dz * dz) >= c->rmin; /* croppers.cpp:149-151 */
      break;
    case ORC_CROP_MIN_MAX_RADIUS: {
      double d = sqrt(dx * dx + dy * dy + dz * dz); /* croppers.cpp:121-124 */
      in = d <= c->rmax && d >= c->rmin;
      break;
    }
    case ORC_CROP_CYLINDER:
      in = p[2] >= c->zmin && p[2] <= c->zmax && sqrt(dx * dx + dy * dy) <= c->rmax; /* croppers.cpp:163-165 */
      break;
    default:
      in = 1; /* CroppingVolume::isWithinVolumeImpl base, croppers.cpp:49-51 */
  }
  return c->invert ? !in : in;
}

size_t orc_crop_indices(const double* pts, size_t n, const orc_crop* c, int64_t* out_idx) {
  size_t k = 0;
  for (size_t i = 0; i < n; ++i)
    if (within_volume(pts + 3 * i, c)) {
      if (out_idx) out_idx[k] = (int64_t)i;
      ++k;
    }
  return k;
}

/* ------------------------------------------------------------------ map merge: src/helpers.cpp:115-183 */
size_t orc_voxelize_within_volume(const double* pts, const double* nrm, size_t n, double voxel, const orc_crop* c, double* out_pts,
                                  double* out_nrm, size_t* n_pass) {
  if (voxel <= 0.0) { /* helpers.cpp:119-123 */
    memcpy(out_pts, pts, sizeof(double) * 3 * n);
    if (nrm && out_nrm) memcpy(out_nrm, nrm, sizeof(double) * 3 * n);
    if (n_pass) *n_pass = n;
    return n;
  }
  const double inv = 1.0 / voxel; /* fromVoxelSize, VoxelHashMap.hpp:43-45 */
  vhash h;
  vhash_init(&h, n);
  double* sp = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  double* sn = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  int64_t* cnt = (int64_t*)calloc(n ? n : 1, sizeof(int64_t));
  size_t m = 0, np = 0;
  for (size_t i = 0; i < n; ++i) {
    const double* p = pts + 3 * i;
    if (within_volume(p, c)) {
      int32_t k[3];
      for (int d = 0; d < 3; ++d) k[d] = (int32_t)floor(p[d] * inv); /* getVoxelIdx, VoxelHashMap.hpp:47-50 */
      int is_new;
      int64_t s = vhash_get(&h, k[0], k[1], k[2], (int64_t)m, &is_new);
      if (is_new) {
        for (int d = 0; d < 3; ++d) sp[3 * m + d] = sn[3 * m + d] = 0.0;
        ++m;
      }
      for (int d = 0; d < 3; ++d) sp[3 * (size_t)s + d] += p[d];
      if (nrm) { /* AccumulatedPoint::AddPoint skips NaN normals, helpers.cpp:34-38 */
        const double* q = nrm + 3 * i;
        if (!isnan(q[0]) && !isnan(q[1]) && !isnan(q[2]))
          for (int d = 0; d < 3; ++d) sn[3 * (size_t)s + d] += q[d];
      }
      cnt[s] += 1;
    } else { /* pass-through, helpers.cpp:155-166 */
      for (int d = 0; d < 3; ++d) {
        out_pts[3 * np + d] = p[d];
        if (nrm && out_nrm) out_nrm[3 * np + d] = nrm[3 * i + d];
      }
      ++np;
    }
  }
  for (size_t s = 0; s < m; ++s) {
    double a[3];
    for (int d = 0; d < 3; ++d) {
      out_pts[3 * (np + s) + d] = sp[3 * s + d] / (double)cnt[s];
      a[d] = sn[3 * s + d] / (double)cnt[s];
    }
    if (nrm && out_nrm) { /* GetAverageNormal().normalized(), helpers.cpp:172 (Eigen: unchanged if norm==0) */
      double z = dot3(a, a);
      if (z > 0.0) {
        double s2 = sqrt(z);
        a[0] /= s2;
        a[1] /= s2;
        a[2] /= s2;
      }
      for (int d = 0; d < 3; ++d) out_nrm[3 * (np + s) + d] = a[d];
    }
  }
  free(sp);
  free(sn);
  free(cnt);
  vhash_free(&h);
  if (n_pass) *n_pass = np;
  return np + m;
}

/* The colours of the same merge (helpers.cpp:40-42,61-63,155-181): out_col is ordered like out_pts of orc_voxelize_within_volume
 * (pass-through points first, then the voxels in first-occurrence order).  AccumulatedPoint::AddPoint ASSIGNS `color_ =
 * cloud.colors_[index]` whenever isValidColor holds, and isValidColor (helpers.cpp:83-85: `c.array().all() >= 0.0 && ... <= 1.0`,
 * a bool compared with a double) holds for every colour, so a voxel carries the colour of its LAST point in cloud order and
 * GetAverageColor returns it undivided. */
size_t orc_voxelize_within_volume_colors(const double* pts, const double* col, size_t n, double voxel, const orc_crop* c, double* out_col) {
  if (voxel <= 0.0) {
    memcpy(out_col, col, sizeof(double) * 3 * n);
    return n;
  }
  const double inv = 1.0 / voxel;
  vhash h;
  vhash_init(&h, n);
  double* last = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  size_t m = 0, np = 0;
  for (size_t i = 0; i < n; ++i) {
    const double* p = pts + 3 * i;
    if (within_volume(p, c)) {
      int32_t k[3];
      for (int d = 0; d < 3; ++d) k[d] = (int32_t)floor(p[d] * inv);
      int is_new;
      int64_t s = vhash_get(&h, k[0], k[1], k[2], (int64_t)m, &is_new);
      if (is_new) ++m;
      for (int d = 0; d < 3; ++d) last[3 * (size_t)s + d] = col[3 * i + d];
    } else {
      for (int d = 0; d < 3; ++d) out_col[3 * np + d] = col[3 * i + d];
      ++np;
    }
  }
  memcpy(out_col + 3 * np, last, sizeof(double) * 3 * m);
  free(last);
  vhash_free(&h);
  return np + m;
}
