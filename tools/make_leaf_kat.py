"""Hand-derived known-answer vectors for the diffusers leaf formulas that oracle/leaf.py restates (the half of the oracle no
reference fixture can pin: diffusers is absent from /root/reference and from this image).

Every value below is computed from the PUBLISHED formula with plain Python ``math`` / ``decimal`` arithmetic (no torch, no
numpy, no import from oracle/ or orv_amd/), or is a closed-form consequence written out in the comment next to it, so that it
is independent of both implementations it checks (tests/test_leaf_kat.py: oracle/leaf.py AND the product's host tables).

    python tools/make_leaf_kat.py          -> tests/golden/leaf_kat.json
"""
import json
import math
import os
from decimal import Decimal, getcontext

getcontext().prec = 60
D = Decimal
kat = {}

# ---------------------------------------------------------------------------------------------------------------------------
# 1. Noise schedule (CogVideoX scheduler_config: 1000 steps, beta 0.00085 -> 0.012 "scaled_linear", snr_shift_scale 3.0,
#    rescale_betas_zero_snr).  Published algorithm (diffusers CogVideoXDDIMScheduler.__init__):
#       beta_i   = (sqrt(b0) + i (sqrt(b1) - sqrt(b0)) / 999)^2                    i = 0..999
#       abar_i   = prod_{j<=i} (1 - beta_j)
#       abar_i  <- abar_i / (s + (1 - s) abar_i),  s = 3                            SNR shift
#       r_i      = sqrt(abar_i);  r_i <- (r_i - r_999) r_0 / (r_0 - r_999);  abar_i <- r_i^2      zero terminal SNR
#    Consequences that need no arithmetic: abar_999 = 0 EXACTLY; r_0 is a fixed point of the rescale, so
#       abar_0 = (1 - b0) / (3 - 2 (1 - b0)).
# ---------------------------------------------------------------------------------------------------------------------------
b0, b1, n, s = D("0.00085"), D("0.012"), 1000, D(3)
sb0, sb1 = b0.sqrt(), b1.sqrt()
abar, acc = [], D(1)
for i in range(n):
    beta = (sb0 + (sb1 - sb0) * D(i) / D(n - 1)) ** 2
    acc *= (1 - beta)
    abar.append(acc)
abar = [a / (s + (1 - s) * a) for a in abar]
r = [a.sqrt() for a in abar]
r0, rT = r[0], r[-1]
abar = [((x - rT) * r0 / (r0 - rT)) ** 2 for x in r]
assert abar[-1] == 0
assert abs(abar[0] - (1 - b0) / (3 - 2 * (1 - b0))) < D("1e-50")
kat["alphas_cumprod"] = {str(t): float(abar[t]) for t in (0, 1, 19, 259, 499, 979, 998, 999)}
kat["alphas_cumprod_closed_form_t0"] = float((1 - b0) / (3 - 2 * (1 - b0)))

# 2. "trailing" timestep spacing: t_k = round(1000 - k * 1000 / N) - 1
kat["trailing_timesteps"] = {str(N): [int(round(1000 - k * 1000 / N)) - 1 for k in range(N)] for N in (50, 4, 3)}
assert kat["trailing_timesteps"]["50"] == list(range(999, 0, -20)) and kat["trailing_timesteps"]["4"] == [999, 749, 499, 249]

# 3. DDIM step (v-prediction, eta = 0), N = 50:  x0 = sqrt(a) x - sqrt(1-a) v ;  A = sqrt((1-ap)/(1-a)) ;
#    x_prev = A x + (sqrt(ap) - sqrt(a) A) x0.
#    * first step t = 999: a = 0  =>  x0 = -v,  A = sqrt(1 - ap),  x_prev = sqrt(1-ap) x - sqrt(ap) v       (ap = abar_979)
#    * last step  t = 19 : prev = -1 => ap = 1 (set_alpha_to_one)  =>  A = 0,  x_prev = x0
def ddim(t, N, x, v):
    a = abar[t]
    tp = t - 1000 // N
    ap = abar[tp] if tp >= 0 else D(1)
    x0 = a.sqrt() * x - (1 - a).sqrt() * v
    A = ((1 - ap) / (1 - a)).sqrt()
    return float(A * x + (ap.sqrt() - a.sqrt() * A) * x0), float(x0)
xv = (D("0.75"), D("-1.25"))
kat["ddim"] = {"x": float(xv[0]), "v": float(xv[1]), "N": 50,
               "t999": ddim(999, 50, *xv), "t499": ddim(499, 50, *xv), "t19": ddim(19, 50, *xv)}
ap = abar[979]
assert abs(D(kat["ddim"]["t999"][0]) - ((1 - ap).sqrt() * xv[0] - ap.sqrt() * xv[1])) < D("1e-15")
assert kat["ddim"]["t19"][0] == kat["ddim"]["t19"][1]

# 4. DPM-Solver++(2M) SDE step (diffusers CogVideoXDPMScheduler.step), N = 50, v-prediction:
#       lam = log sqrt(a/(1-a)), lam' = log sqrt(ap/(1-ap)), h = lam' - lam
#       m1 = sqrt((1-ap)/(1-a)) e^{-h},  m2 = expm1(-2h) sqrt(ap),  mn = sqrt(1-ap) sqrt(1 - e^{-2h})
#       first order : x_prev = m1 x - m2 x0 + mn z
#       second order: r = (lam - lam_back)/h,  d = (1 + 1/(2r)) x0 - (1/(2r)) x0_old,  x_prev = m1 x - m2 d + mn z2
#    * first step t = 999: a = 0 => lam = -inf, h = +inf => m1 = 0, m2 = -sqrt(ap), mn = sqrt(1-ap):
#           x_prev = sqrt(ap) x0 + sqrt(1-ap) z   with x0 = -v
def lam(a):
    return float((a / (1 - a)).sqrt().ln())
def dpm_coeff(t, tb, N):
    a, tp = abar[t], t - 1000 // N
    ap = abar[tp] if tp >= 0 else D(1)
    l, lp = lam(a), lam(ap)
    h = lp - l
    m1 = math.sqrt(float((1 - ap) / (1 - a))) * math.exp(-h)
    m2 = math.expm1(-2 * h) * math.sqrt(float(ap))
    mn = math.sqrt(float(1 - ap)) * math.sqrt(1 - math.exp(-2 * h))
    out = {"m1": m1, "m2": m2, "mn": mn}
    if tb is not None:
        rr = (l - lam(abar[tb])) / h
        out.update(m3=1 + 1 / (2 * rr), m4=1 / (2 * rr))
    return out
kat["dpm"] = {"N": 50, "t979_back999": dpm_coeff(979, None, 50), "t499_back519": dpm_coeff(499, 519, 50),
              "t999_first": {"m1": 0.0, "m2": -math.sqrt(float(abar[979])), "mn": math.sqrt(float(1 - abar[979]))}}

# 5. Timesteps(num_channels=8, flip_sin_to_cos=True, downscale_freq_shift=0): freq_k = exp(-ln(1e4) k / 4) = 10^-k;
#    embedding = [cos(t f_0..3) | sin(t f_0..3)]
def temb(t, dim):
    half = dim // 2
    f = [math.exp(-math.log(10000.0) * k / half) for k in range(half)]
    return [math.cos(t * x) for x in f] + [math.sin(t * x) for x in f]
kat["timestep_embedding"] = {"dim": 8, "t0": temb(0, 8), "t1": temb(1, 8), "t999": temb(999, 8)}
assert kat["timestep_embedding"]["t0"] == [1.0] * 4 + [0.0] * 4

# 6. get_3d_sincos_pos_embed(D=16, (gw=3, gh=2), T=2, spatial scale 1.875, temporal scale 1.0): row = (t, h, w) row-major,
#    features = [temporal D/4 = (sin 2 | cos 2)] [w-coordinate 3D/8 = (sin 3 | cos 3)] [h-coordinate 3D/8 = (sin 3 | cos 3)],
#    omega_k = 10000^(-k / (d/2)) within each group of d features.
def sincos1d(d, p):
    om = [1.0 / 10000 ** (k / (d / 2.0)) for k in range(d // 2)]
    return [math.sin(p * o) for o in om] + [math.cos(p * o) for o in om]
def sincos3d_row(t, h, w, Dm=16, ss=1.875, ts=1.0):
    return sincos1d(Dm // 4, t / ts) + sincos1d(3 * Dm // 8, w / ss) + sincos1d(3 * Dm // 8, h / ss)
kat["sincos3d"] = {"D": 16, "gw": 3, "gh": 2, "T": 2, "spatial_scale": 1.875,
                   "rows": {f"{t},{h},{w}": sincos3d_row(t, h, w) for (t, h, w) in [(0, 0, 0), (1, 0, 0), (0, 0, 2), (0, 1, 0), (1, 1, 2)]}}
assert kat["sincos3d"]["rows"]["0,0,0"] == [0, 0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0, 1, 1, 1]

# 7. 3-D RoPE (get_3d_rotary_pos_embed, head_dim 64 -> 16 | 24 | 24 features for t | h | w, 'linspace' grid with crops
#    ((0,0),(gh,gw)) => positions arange): angle of feature pair i of an axis with d features = pos * 10000^(-2i/d); cos / sin are
#    repeated for the two members of a pair; apply_rotary_emb rotates pairs (x_2i, x_2i+1) -> (x_2i c - x_2i+1 s, x_2i+1 c + x_2i s).
def rope_row(t, h, w):
    cos, sin = [], []
    for d, p in ((16, t), (24, h), (24, w)):
        for i in range(d // 2):
            ang = p * 10000 ** (-2.0 * i / d)
            cos += [math.cos(ang)] * 2
            sin += [math.sin(ang)] * 2
    return cos, sin
kat["rope3d"] = {"head_dim": 64, "gh": 2, "gw": 3, "T": 2,
                 "rows": {f"{t},{h},{w}": rope_row(t, h, w) for (t, h, w) in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 2)]}}
c, sn = rope_row(1, 1, 2)
x = [((7 * i) % 5 - 2) * 0.5 for i in range(64)]
kat["rope_apply"] = {"row": "1,1,2", "x": x,
                     "y": [x[2 * (i // 2)] * c[i] - x[2 * (i // 2) + 1] * sn[i] if i % 2 == 0 else x[i] * c[i] + x[i - 1] * sn[i]
                           for i in range(64)]}
assert abs(sum(v * v for v in kat["rope_apply"]["y"]) - sum(v * v for v in x)) < 1e-12        # rotations preserve the norm

# 8. GELU(tanh): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)));  DiagonalGaussian: std = exp(0.5 clamp(logvar, -30, 20))
kat["gelu_tanh"] = {str(v): 0.5 * v * (1 + math.tanh(math.sqrt(2 / math.pi) * (v + 0.044715 * v ** 3))) for v in (-3.0, -0.5, 0.0, 1.0, 2.5)}
kat["diag_gauss"] = {"mean": 0.25, "eps": -1.5, "logvar": [-40.0, -2.0, 0.0, 25.0],
                     "sample": [0.25 + math.exp(0.5 * min(max(lv, -30.0), 20.0)) * -1.5 for lv in (-40.0, -2.0, 0.0, 25.0)]}

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "leaf_kat.json")
with open(out, "w") as f:
    json.dump(kat, f, indent=1)
print("wrote", out)
