"""CU split (dlv_set_cu_split): the convs of a forward on the large CU partition, its HBM-class kernels on the small one, on
CU-masked streams with an event per hop.  Same kernels in the same order per window: the pass must be bit-identical to the
unsplit one, for every split and lane count, with and without the count map, also when a batch is smaller than the default."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_cu_split_pass_is_bit_identical(prec):
    import torch
    from delivr_cfos_amd.engine import HipEngine
    from delivr_cfos_amd.synth import synth_volume_np
    from delivr_cfos_amd.weights import random_state_dict

    eng = HipEngine(0)
    eng.load_state_dict({"state_dict": random_state_dict(3)})
    shape, roi = (128, 256, 384), (128, 128, 128)
    vol = synth_volume_np(shape, seed=8)
    vol[:, :, 250:] = 0  # background windows: the skip fill runs on the main stream next to the lanes
    v = eng.to_device(vol)

    def run(split, lanes, flip=None):
        eng.set_cu_split(split)
        eng.set_lanes(lanes)
        acc = torch.zeros(shape, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(shape, dtype=torch.uint8, device="cuda")
        st = eng.sw_infer(eng.make_sw_params(shape, roi, 0.5, flip, 0, prec), v, acc, cnt)
        eng.sync()
        return acc.cpu().numpy(), cnt.cpu().numpy(), st

    base, cbase, st0 = run(0, 3)
    assert st0["n_windows"] == 15 and st0["n_skipped"] > 0
    for split, lanes in ((8, 1), (8, 3), (6, 4), (12, 6), (-1, 3)):
        a, c, st = run(split, lanes)
        assert st["n_windows"] == st0["n_windows"] and st["n_skipped"] == st0["n_skipped"]
        np.testing.assert_array_equal(c, cbase)
        np.testing.assert_array_equal(a, base, err_msg=f"split {split}, lanes {lanes}")
    fb, _, _ = run(0, 3, flip=3)
    fs, _, _ = run(8, 3, flip=3)
    np.testing.assert_array_equal(fs, fb)
    eng.set_cu_split(0)
    eng.close()
