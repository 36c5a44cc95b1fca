// Host build of the product's scene generator (eb-cadrl_amd/csrc/ebc_scene_gen.h) for the CPU tests: the same
// source the device kernel compiles, run scene by scene, so tests/test_scene_gen.py can hold it against
// ebcsim/scene.py (= the reference's SceneGenerator, tests/golden/scenes.json) without a GPU.
#include <vector>
#include "../../eb-cadrl_amd/csrc/ebc_scene_gen.h"

extern "C" int scene_gen_host(const EbcSceneGen *gen, const uint32_t *seeds, int n, int N, int S, int G, int *n_humans,
                              double *px, double *py, double *vx, double *vy, double *gx, double *gy, double *radius,
                              double *v_pref, uint8_t *type, int *n_static, double *spx, double *spy, double *sradius,
                              uint64_t *grid, double *robot) {
  std::vector<uint32_t> mt(624);
  int status = 0;
  const int S1 = S ? S : 1;
  for (int r = 0; r < n; ++r) {
    const size_t k = (size_t)r * N, q = (size_t)r * S1;
    ebc::SceneRow o = {n_humans + r, px + k, py + k, vx + k, vy + k, gx + k, gy + k, radius + k, v_pref + k, type + k,
                       n_static + r, spx + q, spy + q, sradius + q, grid ? grid + (size_t)r * G * 2 : nullptr,
                       robot + (size_t)r * 9};
    status |= ebc::generate_scene_row(*gen, seeds[r], mt.data(), 1, o, N, S, G);
  }
  return status;
}

extern "C" void sincos_host(const double *angle, int n, double *c, double *s) {
  for (int i = 0; i < n; ++i) ebc::sincos_dd(angle[i], c[i], s[i]);
}
