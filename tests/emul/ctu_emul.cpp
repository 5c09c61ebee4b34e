// CPU emulation of uvg266_amd/csrc/ctu_core.h for the "-m not gpu" tests: the kernel's source compiled for the host, one lane
// playing the whole workgroup (PAR_FOR = a plain loop, no barriers).  It checks the LOGIC of the device code against the oracle
// where there is no GPU; it is test infrastructure (built by tests/emul/Makefile into tests/emul/_build/), never linked into
// libuvg266hip.so, and nothing in the product can reach it.
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include "ctu_core.h"

template <typename PX>
static int run_picture(const ctu::params &P, const PX *sy, const PX *su, const PX *sv, PX *ry, PX *ru, PX *rv, uvghip_scu_t *cu_tab, int16_t *coeff,
                       uint32_t *models)
{
  const int W = P.pic_w, H = P.pic_h, wc = (W + 63) / 64, hc = (H + 63) / 64;
  ctu::lds<PX> *S = new ctu::lds<PX>;
  ctu::scratch *Wk = new ctu::scratch;
  for (int cy = 0; cy < hc; ++cy)
    for (int cx = 0; cx < wc; ++cx) {
      memset(S, 0xA5, sizeof *S);          // nothing may depend on what the LDS or the global scratch held before
      memset(Wk, 0xA5, sizeof *Wk);
      ctu::job<PX> J;
      J.P = P;
      J.src_y = sy; J.src_u = su; J.src_v = sv; J.src_stride = W; J.src_stride_c = W / 2;
      J.rec_y = ry; J.rec_u = ru; J.rec_v = rv; J.rec_stride = W; J.rec_stride_c = W / 2;
      J.cu_tab = cu_tab; J.cu_stride = wc * 16;
      const int k = cy * wc + cx;
      J.coeff = coeff + (size_t)k * 6144;
      J.models_out = models + (size_t)k * 3 * ctu::NMODELS;
      J.models_in = cx > 0 ? models + ((size_t)(k - 1) * 3 + 2) * ctu::NMODELS : (cy > 0 ? models + ((size_t)((cy - 1) * wc) * 3 + 2) * ctu::NMODELS : nullptr);
      J.W = Wk;
      J.x = cx * 64; J.y = cy * 64;
      ctu::run_ctu(S, J);
    }
  delete S; delete Wk;
  return 0;
}

extern "C" __attribute__((visibility("default"))) void ctu_emul_set_lazy(int on) { ctu::g_emul_lazy = on; }

extern "C" __attribute__((visibility("default")))
int ctu_emul_search_picture(int bitdepth, const ctu::params *P, const void *sy, const void *su, const void *sv, void *ry, void *ru, void *rv,
                            uvghip_scu_t *cu_tab, int16_t *coeff, uint32_t *models)
{
  if (bitdepth == 8) return run_picture<uint8_t>(*P, (const uint8_t *)sy, (const uint8_t *)su, (const uint8_t *)sv, (uint8_t *)ry, (uint8_t *)ru, (uint8_t *)rv, cu_tab, coeff, models);
  if (bitdepth == 10) return run_picture<uint16_t>(*P, (const uint16_t *)sy, (const uint16_t *)su, (const uint16_t *)sv, (uint16_t *)ry, (uint16_t *)ru, (uint16_t *)rv, cu_tab, coeff, models);
  return -1;
}
