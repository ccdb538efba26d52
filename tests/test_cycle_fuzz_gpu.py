"""One full optimisation cycle (nine terms), the one-euro filters, and a second cycle with the filtered-vertex term, on
random small sequences against the CPU oracle: frame counts that are no batch multiples (ragged last batch, batch larger than
the sequence), 1-3 humans, landscape / portrait / square images, with and without a scene cloud.  EVERY entry of every leaf
gradient and every log entry (deterministic scatter; the oracle renders the faces the kernel selected at the vertices the
kernel produced, as in tests/test_full_size_gpu.py -- with its renderer in float32 AND in float64; an entry is held against
float64 first; the few that miss it must agree with float32 and are counted and bounded (tests/parity_gates.py;
test_fit_full_gpu._oracle_grads_both says why neither precision is the truth for every entry).  tools/fuzz_cycle.py
runs the same loop for any number of sequences: 180 of them (single frames, up to six humans, 32x24 to 200x40 among them)
had 5.7e-5 as the worst entry of the five large leaves, 1.6e-4 on the scale leaf and 1.3e-6 on the log; three sequences
needed float32 (float64 up to 6.7e-4 off), one needed float64 (float32 2.8e-3 off)."""
import numpy as np
import pytest

from parity_gates import two_precision_gate

from mhhip import synthetic
import test_fit_full_gpu as tf
from test_full_size_gpu import LOG_KEYS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed', [515, 626])
def test_random_sequences_cycle_filters_cycle(smpl_struct, smpl_regs, oracle_model, tmp_path, seed):
    from mhhip.raster import RasterTerms, set_deterministic
    rng = np.random.RandomState(seed)
    second = {}
    old = set_deterministic(True)
    try:
        for c in range(4):
            W, H = [(96, 54), (64, 96), (80, 80), (120, 68)][rng.randint(4)]
            T, N = int(rng.randint(3, 14)), int(rng.randint(1, 4))
            batch = int(rng.choice([2, 3, 5, 7]))
            scene = bool(rng.randint(2))
            sub = tmp_path / ('c%d' % c)
            sub.mkdir()
            opt, dl, o, batches, seq = tf._setup(smpl_struct, smpl_regs, oracle_model, sub, T, N, W, H, batch, int(rng.randint(1 << 30)), scene)
            opt._stage_from_dataloader(dl)
            e = opt.engine
            raster = RasterTerms(e)
            hsel = tf._HipSelectionRasteriser(np.asarray(smpl_struct.f).astype(np.int64), synthetic.default_cam_K((W, H), 60.0), (W, H), N)
            o.rasteriser = hsel
            where = 'sequence %d (%dx%d, T %d, N %d, batch %d, scene %s)' % (c, W, H, T, N, batch, scene)
            for cyc in range(2):
                if cyc == 1:
                    e.update_filters()
                    o.update_filters()
                e.cycle(cyc, raster=raster)
                hsel.take(raster, e, oracle=o)
                log = e.read_log(cyc + 1)[cyc]
                want, both = tf._oracle_grads_both(o, hsel, batches)
                for k in LOG_KEYS + (['reg_filter_verts'] if cyc else []):
                    np.testing.assert_allclose(log[k], want[k], rtol=2e-5, atol=1e-7, err_msg='%s cycle %d %s' % (where, cyc, k))
                for name, ename in tf.LEAF_MAP:
                    w32, w64 = both[name]
                    g = e.leaf(ename, e.grads).cpu().numpy().reshape(w32.shape)
                    # xscale: N entries, each the sum of everything a person's vertices receive -- it can cancel to ~0
                    tol = 1e-3 if name == 'xscale' else 1e-4
                    # (one edge decision of the rasteriser reaches every pose entry of its body: the cap is per leaf, 2 %)
                    worst, n32 = two_precision_gate(g, w32, w64, tol, '%s cycle %d leaf %s' % (where, cyc, name),
                                                    max_second=max(3, int(0.02 * g.size)))
                    second[name] = second.get(name, 0) + n32
    finally:
        set_deterministic(old)
    print('entries that needed the float32 oracle, by leaf:', second)
