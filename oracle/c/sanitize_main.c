/*
 * Sanitizer driver of the plain-C oracle (TEST INFRASTRUCTURE ONLY; SURVEY §5 "race detection / sanitizers": the GPU
 * pool has no GPU AddressSanitizer, so the CPU restatement is what runs under ASan/UBSan).
 * Builds a seeded scene that drives the rare branches (empty tiles, partial last tile row/column, Gaussians behind the
 * camera, outside the frustum, huge and sub-pixel splats, SH degrees 0..3, colours / covariances precomputed), runs
 * gsr_oracle_forward in both of its modes (count, then fill) and prints a checksum the plain build must reproduce.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

long gsr_oracle_forward(int P, int M, int D, int W, int H, float tanfovx, float tanfovy, float scale_modifier,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, const float* rotations, const float* cov3D_precomp, const float* V,
                        const float* Mx, const float* campos, const float* bg, float* out_color, int32_t* radii,
                        float* final_T, uint32_t* n_contrib, uint64_t* keys_out, uint32_t* list_out, int64_t* ranges_out);

static uint64_t state = 88172645463325252ull;
static float urand(void) {   /* xorshift64, [0,1) */
  state ^= state << 13; state ^= state >> 7; state ^= state << 17;
  return (float)((state >> 40) & 0xffffff) / 16777216.0f;
}
static float nrand(void) { return sqrtf(-2.0f * logf(urand() + 1e-7f)) * cosf(6.2831853f * urand()); }

static double run(int P, int deg, int W, int H, int use_cov, int use_col) {
  const int M = (deg + 1) * (deg + 1);
  float* xyz = malloc(sizeof(float) * 3 * P); float* sh = malloc(sizeof(float) * 3 * M * P);
  float* col = malloc(sizeof(float) * 3 * P); float* op = malloc(sizeof(float) * P);
  float* sc = malloc(sizeof(float) * 3 * P); float* rot = malloc(sizeof(float) * 4 * P); float* cov = malloc(sizeof(float) * 6 * P);
  for (int i = 0; i < P; ++i) {
    xyz[3 * i] = 6.0f * (2 * urand() - 1); xyz[3 * i + 1] = 3.4f * (2 * urand() - 1);
    xyz[3 * i + 2] = (i % 11 == 0) ? -2.0f : (i % 13 == 0 ? 0.2f : 3.0f + 6.0f * urand());   /* behind / on the near plane */
    const float ls = (i % 7 == 0) ? 1.0f : (i % 5 == 0 ? -7.0f : -3.0f);                       /* huge / sub-pixel / normal */
    float q[4], n = 0;
    for (int k = 0; k < 3; ++k) sc[3 * i + k] = expf(ls + 0.4f * nrand());
    for (int k = 0; k < 4; ++k) { q[k] = nrand(); n += q[k] * q[k]; }
    for (int k = 0; k < 4; ++k) rot[4 * i + k] = q[k] / sqrtf(n + 1e-12f);
    op[i] = 1.0f / (1.0f + expf(-1.5f * nrand()));
    for (int k = 0; k < 3 * M; ++k) sh[(size_t)3 * M * i + k] = (k < 3 ? 1.0f : 0.3f) * nrand();
    for (int k = 0; k < 3; ++k) col[3 * i + k] = urand();
    /* covariance = diag(s^2) (axis aligned): enough for the precomputed-covariance path */
    cov[6 * i] = sc[3 * i] * sc[3 * i]; cov[6 * i + 1] = 0; cov[6 * i + 2] = 0;
    cov[6 * i + 3] = sc[3 * i + 1] * sc[3 * i + 1]; cov[6 * i + 4] = 0; cov[6 * i + 5] = sc[3 * i + 2] * sc[3 * i + 2];
  }
  const float fx = 0.6f * W, fy = 0.6f * W, tanx = W / (2 * fx), tany = H / (2 * fy), zn = 0.01f, zf = 100.0f;
  float V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float Pm[16] = {0};   /* row-vector convention: transposed getProjectionMatrix, times the identity view */
  Pm[0] = 1.0f / tanx; Pm[5] = 1.0f / tany; Pm[10] = zf / (zf - zn); Pm[11] = 1.0f; Pm[14] = -(zf * zn) / (zf - zn);
  float campos[3] = {0, 0, 0}, bg[3] = {0.1f, 0.2f, 0.3f};
  float* out = calloc((size_t)3 * W * H, sizeof(float)); int32_t* radii = calloc(P, sizeof(int32_t));
  float* fT = calloc((size_t)W * H, sizeof(float)); uint32_t* nc = calloc((size_t)W * H, sizeof(uint32_t));
  const int T = ((W + 15) / 16) * ((H + 15) / 16);
  long R = gsr_oracle_forward(P, use_col ? 0 : M, deg, W, H, tanx, tany, 1.0f, xyz, use_col ? NULL : sh, use_col ? col : NULL, op,
                              use_cov ? NULL : sc, use_cov ? NULL : rot, use_cov ? cov : NULL, V, Pm, campos, bg, out, radii,
                              fT, nc, NULL, NULL, NULL);
  if (R < 0) { fprintf(stderr, "oracle failed\n"); exit(2); }
  uint64_t* keys = malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1)); uint32_t* list = malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
  int64_t* ranges = calloc((size_t)2 * T, sizeof(int64_t));
  long R2 = gsr_oracle_forward(P, use_col ? 0 : M, deg, W, H, tanx, tany, 1.0f, xyz, use_col ? NULL : sh, use_col ? col : NULL, op,
                               use_cov ? NULL : sc, use_cov ? NULL : rot, use_cov ? cov : NULL, V, Pm, campos, bg, out, radii,
                               fT, nc, keys, list, ranges);
  if (R2 != R) { fprintf(stderr, "count / fill disagree\n"); exit(3); }
  double sum = (double)R;
  for (size_t i = 0; i < (size_t)3 * W * H; ++i) sum += out[i];
  for (long i = 0; i < R; ++i) sum += (double)(list[i] % 97) + (double)(keys[i] >> 32);
  for (int i = 0; i < P; ++i) sum += radii[i];
  for (int i = 0; i < 2 * T; ++i) sum += (double)ranges[i];
  free(xyz); free(sh); free(col); free(op); free(sc); free(rot); free(cov); free(out); free(radii); free(fT); free(nc);
  free(keys); free(list); free(ranges);
  return sum;
}

int main(void) {
  double total = 0;
  total += run(1500, 3, 211, 117, 0, 0);     /* partial last tile row and column */
  total += run(700, 0, 64, 48, 1, 1);        /* precomputed colours and covariances */
  total += run(300, 1, 17, 33, 0, 0);
  total += run(0, 2, 40, 40, 0, 0);          /* no Gaussians at all */
  total += run(5, 2, 16, 16, 0, 1);
  printf("%.6f\n", total);
  return 0;
}
