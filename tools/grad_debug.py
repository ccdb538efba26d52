"""developer tool: for one random scene of tools/fuzz_raster_grads.py (SEED, CASE), the entries of dL/dverts where the HIP
kernel and the float32 oracle disagree, against the float64 oracle: whose error is it?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd'), os.path.join(ROOT, 'tests')]
from mhhip import synthetic
import test_raster_gpu as tr
struct = synthetic.make_smpl_struct(1); regs = synthetic.make_extra_regressors(1, struct)
rng = np.random.RandomState(int(os.environ.get('SEED', '7')))
want_case = int(os.environ.get('CASE', '23'))
for c in range(want_case + 1):
    W, H = [(96, 54), (64, 96), (80, 80), (160, 90), (48, 135), (240, 135)][rng.randint(6)]
    T, N = int(rng.randint(1, 3)), int(rng.randint(1, 4))
    zlo = float(rng.choice([1.1, 1.6, 2.5, 4.0]))
    zhi = zlo + float(rng.choice([0.3, 1.0, 3.0]))
    fov = float(rng.choice([40.0, 60.0, 90.0]))
    seed = int(rng.randint(1 << 30))
kw = dict(zlo=zlo, zhi=zhi, fov=fov, hip_selection=True)
if os.environ.get('TERM') == 'depth': kw['coefs'] = dict(depth=0.05, silhouette=0.0)
if os.environ.get('TERM') == 'sil': kw['coefs'] = dict(depth=0.0, silhouette=0.1)
r32 = tr._run_case(struct, regs, T, N, W, H, seed, **kw)
r64 = tr._run_case(struct, regs, T, N, W, H, seed, oracle_dtype=torch.float64, **kw)
g, w32, w64 = r32['gv'].astype(np.float64), r32['want_gv'].astype(np.float64), r64['want_gv']
assert np.array_equal(r32['gv'], r64['gv']) or True
scale = np.abs(w64).max()
e_hip, e_o32 = np.abs(g - w64) / scale, np.abs(w32 - w64) / scale
print('scene %dx%d T%d N%d z %.1f-%.1f fov %.0f; scale %.3e' % (W, H, T, N, zlo, zhi, fov, scale))
print('HIP vs float64 oracle:     max %.2e, entries > 2e-4: %d' % (e_hip.max(), int((e_hip > 2e-4).sum())))
print('float32 vs float64 oracle: max %.2e, entries > 2e-4: %d' % (e_o32.max(), int((e_o32 > 2e-4).sum())))
bad = np.argwhere((e_hip > 2e-4) | (e_o32 > 2e-4))
for b, v, c in bad[:24]:
    print('  body %d vertex %4d comp %d: hip %+.6e  o32 %+.6e  o64 %+.6e   hip err %.1e  o32 err %.1e'
          % (b, v, c, g[b, v, c], w32[b, v, c], w64[b, v, c], e_hip[b, v, c], e_o32[b, v, c]))

# ---- which pixels feed the worst entry, and what the depth term's per-pixel NDC gradient is in float32 / float64 ----------
from oracle import raster_oracle as ro
b, v, c = [int(q) for q in bad[np.argmax([e_hip[tuple(q)] for q in bad])]]
print('worst HIP entry: body %d vertex %d comp %d' % (b, v, c))
T_, N_, H_, W_ = r32['shape']
faces = r32['faces']
got = tr._hip_selection(r32['sel'], T_ * N_, H_, W_)
xs, ys = ro.pixel_centres_ndc(H_, W_)
fs = np.nonzero((faces == v).any(axis=1))[0]
ndc64 = ro.to_ndc(torch.tensor(r32['verts']).double(), r32['K'], (W_, H_)).numpy()
ndc32 = ro.to_ndc(torch.tensor(r32['verts']), r32['K'], (W_, H_)).numpy()
print('max |ndc32 - ndc64| of the body: %.2e' % np.abs(ndc32[b] - ndc64[b]).max())

def dpz(ndc, f, xf, yf, dt):
    """d pz / d (ndc x, y, z of the three vertices) of the clipped-barycentric depth, by autograd in dtype dt"""
    t = torch.tensor(ndc[b][faces[f]], dtype=dt, requires_grad=True)
    x0, y0, z0 = t[0]; x1, y1, z1 = t[1]; x2, y2, z2 = t[2]
    edge = lambda px, py, ax, ay, bx, by: (px - ax) * (by - ay) - (py - ay) * (bx - ax)
    px, py = torch.tensor(xf, dtype=dt), torch.tensor(yf, dtype=dt)
    area = edge(x2, y2, x0, y0, x1, y1) + 1e-8
    w = [edge(px, py, x1, y1, x2, y2) / area, edge(px, py, x2, y2, x0, y0) / area, edge(px, py, x0, y0, x1, y1) / area]
    cc = [torch.clamp(q, min=0) for q in w]
    cs = torch.clamp(cc[0] + cc[1] + cc[2], min=1e-5)
    pz = (cc[0] / cs) * z0 + (cc[1] / cs) * z1 + (cc[2] / cs) * z2
    pz.backward()
    return float(pz), t.grad.numpy().astype(np.float64), float(area), [float(q) for q in w]

for y, x in np.argwhere(np.isin(got[b, :, :, 0], fs)):
    f = int(got[b, y, x, 0]); k = int(np.nonzero(faces[f] == v)[0][0])
    p64 = dpz(ndc64, f, float(xs[x]), float(ys[y]), torch.float64)
    p32 = dpz(ndc32, f, float(xs[x]), float(ys[y]), torch.float32)
    p6432 = dpz(ndc32.astype(np.float64), f, float(xs[x]), float(ys[y]), torch.float64)
    print('  pixel (%d,%d) face %d: area %.3e w %s' % (x, y, f, p64[2], ['%.3f' % q for q in p64[3]]))
    print('      dpz/d(x,y,z) of the vertex: f64 %s | f64 at the fp32 ndc %s | f32 %s' % (p64[1][k], p6432[1][k], p32[1][k]))

def emu(ndc, f, xf, yf, F):
    """the kernel's depth-term formulas (rg_pixel) for d pz / d ndc of the three vertices, in arithmetic F"""
    t = ndc[b][faces[f]].astype(F)
    X, Y, Z = t[:, 0], t[:, 1], t[:, 2]
    xf, yf = F(xf), F(yf)
    edge = lambda px, py, ax, ay, bx, by: (px - ax) * (by - ay) - (py - ay) * (bx - ax)
    area = edge(X[2], Y[2], X[0], Y[0], X[1], Y[1]) + F(1e-8)
    ia = F(1) / area
    w = np.array([edge(xf, yf, X[1], Y[1], X[2], Y[2]) * ia, edge(xf, yf, X[2], Y[2], X[0], Y[0]) * ia, edge(xf, yf, X[0], Y[0], X[1], Y[1]) * ia], F)
    cc = np.maximum(w, F(0))
    craw = cc[0] + cc[1] + cc[2]
    cs = max(craw, F(1e-5)); ics = F(1) / cs
    nw = (cc / cs).astype(F)
    gpz = F(1)
    gw = np.zeros(3, F)
    for k in range(3):
        if craw > F(1e-5):
            acc = F(0)
            for j in range(3):
                if j != k: acc = acc + (Z[k] - Z[j]) * nw[j]
            gc = gpz * acc * ics
        else:
            gc = gpz * Z[k] * ics
        gw[k] = gc if w[k] > 0 else F(0)
    ge = (gw * ia).astype(F)
    garea = -(gw[0] * w[0] + gw[1] * w[1] + gw[2] * w[2]) * ia
    gx, gy = np.zeros(3, F), np.zeros(3, F)
    def adj(gE, A, B):
        gx[A] += gE * (yf - Y[B]); gy[A] += gE * (X[B] - xf)
        gx[B] += gE * (-(yf - Y[A])); gy[B] += gE * (xf - X[A])
    adj(ge[0], 1, 2); adj(ge[1], 2, 0); adj(ge[2], 0, 1)
    gx[2] += garea * (Y[1] - Y[0]); gy[2] += garea * (-(X[1] - X[0]))
    gx[0] += garea * (Y[2] - Y[1]); gy[0] += garea * (X[1] - X[2])
    gx[1] += garea * (-(Y[2] - Y[0])); gy[1] += garea * (X[2] - X[0])
    return gx, gy, gpz * nw

print('the kernel\'s formulas for the sliver pixels, float32 / float64 arithmetic on the fp32 NDC coordinates:')
for y, x in np.argwhere(np.isin(got[b, :, :, 0], fs)):
    f = int(got[b, y, x, 0]); k = int(np.nonzero(faces[f] == v)[0][0])
    e32 = emu(ndc32, f, float(xs[x]), float(ys[y]), np.float32)
    e64 = emu(ndc32, f, float(xs[x]), float(ys[y]), np.float64)
    print('  pixel (%d,%d) face %d: formulas f32 (%.7g, %.7g, %.7g) | f64 (%.7g, %.7g, %.7g)'
          % (x, y, f, e32[0][k], e32[1][k], e32[2][k], e64[0][k], e64[1][k], e64[2][k]))
