"""The same reconstruction of the encoder's inter decisions (tests/test_oracle_inter_recon.py) through the DEVICE's block kernels:
uvghip_mc_batch (integer copies, 8-tap / 4-tap fractional samplers, pixels or 14-bit intermediates, border replication),
uvghip_bipred_average_batch, uvghip_dequant_batch + uvghip_transform_batch -- every inter CU of two low-delay encodes of the real
encoder (BASELINE configs[2]: --gop lp-g4d3t1 --preset medium) comes out as the encoder reconstructed it.  This is the decoder
side of the inter path on the GPU; the search side is what comes next."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


class DeviceBlocks:
    """The three operations of helpers.inter_reconstruct on the device (blocks cross to the host as numpy in between)."""
    def __init__(self, depth):
        import torch
        from uvg266_amd import api
        self.t, self.api, self.depth = torch, api, depth

    def up(self, a):
        return self.t.from_numpy(np.ascontiguousarray(a)).cuda()

    def picture(self, planes):
        return tuple(self.up(p) for p in planes)

    def sample(self, plane, pw, ph, x0, y0, w, h, fx, fy, chroma, hi):
        blks = self.t.tensor([[x0, y0, fx, fy]], dtype=self.t.int32, device="cuda")
        return self.api.mc_batch(plane, blks, w, h, pw, ph, is_chroma=chroma, hi=hi)[0].reshape(-1).cpu().numpy()

    def average(self, a, b, w, h):
        return self.api.bipred_average_batch(self.up(a), self.up(b), self.depth).cpu().numpy()

    def residual(self, levels, n, qp, color):
        q = self.up(np.ascontiguousarray(levels, np.int16).reshape(1, n, n))
        qs = qp + 6 * (self.depth - 8)              # uvg_get_scaled_qp with the identity chroma table (transform.c:150-165)
        r = self.api.transform_batch(self.api.dequant_batch(q, self.depth, qs), self.depth, inverse=True)
        return r[0].cpu().numpy().astype(np.int32)


@pytest.mark.parametrize("name", ["ref_inter_192x128_8_qp17_5frames", "ref_inter_136x72_10_qp22_4frames"])
def test_device_kernels_reconstruct_the_encoders_inter_pictures(hip, name):
    g = H.ctu_golden(name)
    seen = H.inter_reconstruct(g, DeviceBlocks(int(g["dims"][2])))
    assert seen["inter"] > 100 and seen["bi"] > 20 and seen["frac"] > 50 and seen["resid"] > 50, seen
