"""A second, independent implementation of the reference's geodesic driver -- plain Python floats, written
from the Rust sources alone and sharing no code with oracle/ (which is C) or with the HIP kernels:

  integrate()             physics-engine/gravitas-core/src/geodesic/mod.rs:180-265
  AdaptiveStepper::step   .../geodesic/integrator.rs:53-108
  adaptive_rkf45_step     .../geodesic/integrator.rs:113-190
  get_state_derivative    .../geodesic/hamiltonian.rs:13-35
  Kerr (BL and KS)        .../metric/kerr.rs:266-499, event_horizon .../metric/mod.rs:75-84
  renormalize_null        .../invariants/renormalization.rs:13-45
  hamiltonian             .../invariants/mod.rs:25-37

tests/test_oracle_independent.py holds the oracle (and through it every bit-exact HIP path) to this file on the
committed ray list tests/golden/rays_v2.npz: equal termination class, equal steps_taken, end states within 1e-6.
The transcendental functions are the host libm's (math.sin / cos / pow), NOT the oracle's specified ones, so the
comparison also measures what a different last bit in sin / cos / pow does along a whole ray.

Deliberately scalar and unoptimised: one `State` tuple in, one out; Rust's NaN conventions for max / min /
clamp / signum are spelled out where the controller depends on them (integrator.rs:76-104)."""
import math

NONE, HORIZON, ESCAPE, MAXSTEPS = 0, 1, 2, 3
BL, KS = 0, 1


def rmax(a, b):
    """f64::max: the other operand when one is NaN"""
    if a != a:
        return b
    if b != b:
        return a
    return a if a > b else b


def rmin(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return a if a < b else b


def rclamp(x, lo, hi):
    """f64::clamp: NaN stays NaN"""
    if x != x:
        return x
    return lo if x < lo else (hi if x > hi else x)


def rsignum(x):
    if x != x:
        return x
    return math.copysign(1.0, x)


def rpow(x, y):
    try:
        return math.pow(x, y)
    except (OverflowError, ValueError, ZeroDivisionError):
        if x != x:
            return x
        if x == 0.0 and y < 0.0:
            return math.inf
        if x < 0.0:
            return math.nan
        return math.inf


class KerrHole:
    """`Kerr::new` / `Kerr::kerr_schild` (kerr.rs:48-63): spin clamped to [-1, 1]; a = a* M (kerr.rs:70-74)"""

    def __init__(self, mass, spin, coords):
        self.M = float(mass)
        self.spin = rclamp(float(spin), -1.0, 1.0)
        self.a = self.spin * self.M
        self.coords = coords

    def horizon(self):  # metric/mod.rs:75-84
        disc = self.M * self.M - self.a * self.a
        return self.M if disc < 0.0 else self.M + math.sqrt(disc)

    # --- inverse metric as the seven entries the driver touches: tt, tr, tphi, rr, rphi, thth, phph ---
    def inverse(self, r, th):
        M, a = self.M, self.a
        r2, a2 = r * r, a * a
        if self.coords == KS:  # kerr.rs:412-440
            s = math.sin(th)
            sin2 = rmax(s * s, 1e-12)
            cos2 = 1.0 - sin2
            sigma = r2 + a2 * cos2
            delta = r2 - 2.0 * M * r + a2
            gtt = -(1.0 + 2.0 * M * r / sigma)
            gtr = 2.0 * M * r / sigma
            return (gtt, gtr, 0.0, delta / sigma, a / sigma, 1.0 / sigma, 1.0 / (sigma * sin2))
        s, c = math.sin(th), math.cos(th)  # kerr.rs:266-292
        sin2, cos2 = s * s, c * c
        sigma = r2 + a2 * cos2
        delta = r2 - 2.0 * M * r + a2
        gtt = -((sigma * (r2 + a2) + 2.0 * M * r * a2 * sin2) / (delta * sigma))
        gphph = 0.0 if sin2 < 1e-9 else (delta - a2 * sin2) / (delta * sigma * sin2)
        gtph = -(2.0 * M * r * a) / (delta * sigma)
        return (gtt, 0.0, gtph, delta / sigma, 0.0, 1.0 / sigma, gphph)

    # --- (dH/dr, dH/dtheta) ---
    def dH(self, r, th, p):
        M, a = self.M, self.a
        r2, a2 = r * r, a * a
        pt, pr, pth, pph = p
        s, c = math.sin(th), math.cos(th)
        if self.coords == KS:  # kerr.rs:442-499
            sin2 = rmax(s * s, 1e-12)
            cos2 = 1.0 - sin2
            sigma = r2 + a2 * cos2
            sigma2 = sigma * sigma
            delta = r2 - 2.0 * M * r + a2
            ds_r = 2.0 * r
            ds_t = -2.0 * a2 * s * c
            dd_r = 2.0 * r - 2.0 * M
            tt_r = -(2.0 * M * (sigma - r * ds_r)) / sigma2
            tt_t = (2.0 * M * r * ds_t) / sigma2
            tr_r, tr_t = -tt_r, -tt_t
            rr_r = (dd_r * sigma - delta * ds_r) / sigma2
            rr_t = -(delta * ds_t) / sigma2
            hh_r = -ds_r / sigma2
            hh_t = -ds_t / sigma2
            ff_r = -ds_r / (sigma2 * sin2)
            ff_t = -(ds_t * sin2 + sigma * 2.0 * s * c) / (sigma2 * sin2 * sin2)
            rf_r = -(a * ds_r) / sigma2
            rf_t = -(a * ds_t) / sigma2
            d_r = 0.5 * (tt_r * pt * pt + rr_r * pr * pr + hh_r * pth * pth + ff_r * pph * pph
                         + 2.0 * tr_r * pt * pr + 2.0 * rf_r * pr * pph)
            d_t = 0.5 * (tt_t * pt * pt + rr_t * pr * pr + hh_t * pth * pth + ff_t * pph * pph
                         + 2.0 * tr_t * pt * pr + 2.0 * rf_t * pr * pph)
            if abs(s) < 1e-10:
                d_t = 0.0
            return d_r, d_t
        # Boyer-Lindquist, kerr.rs:294-388
        sin2, cos2 = s * s, c * c
        sigma = r2 + a2 * cos2
        delta = r2 - 2.0 * M * r + a2
        sigma_sq = sigma * sigma
        ds_r = 2.0 * r
        ds_t = -2.0 * a2 * c * s
        dd_r = 2.0 * r - 2.0 * M
        rr_r = (dd_r * sigma - delta * ds_r) / sigma_sq
        rr_t = -(delta * ds_t) / sigma_sq
        hh_r = -ds_r / sigma_sq
        hh_t = -ds_t / sigma_sq
        num = -2.0 * M * r * a
        den = delta * sigma
        dnum_r = -2.0 * M * a
        dden_r = dd_r * sigma + delta * ds_r
        tf_r = (dnum_r * den - num * dden_r) / (den * den)
        dden_t = delta * ds_t
        tf_t = -(num * dden_t) / (den * den)
        du_r = ds_r * (r2 + a2) + sigma * 2.0 * r + 2.0 * M * a2 * sin2
        u = sigma * (r2 + a2) + 2.0 * M * r * a2 * sin2
        tt_r = -(du_r * den - u * dden_r) / (den * den)
        du_t = ds_t * (r2 + a2) + 2.0 * M * r * a2 * 2.0 * s * c
        tt_t = -(du_t * den - u * dden_t) / (den * den)
        da_r = -ds_r / (sigma_sq * sin2)
        db_r = -a2 * dden_r / (den * den)
        ff_r = da_r - db_r
        dda_t = ds_t * sin2 + sigma * 2.0 * s * c
        da_t = -dda_t / (sigma_sq * sin2 * sin2)
        db_t = -a2 * dden_t / (den * den)
        ff_t = da_t - db_t
        d_r = 0.5 * (pt * pt * tt_r + pr * pr * rr_r + pth * pth * hh_r + pph * pph * ff_r + 2.0 * pt * pph * tf_r)
        d_t = 0.5 * (pt * pt * tt_t + pr * pr * rr_t + pth * pth * hh_t + pph * pph * ff_t + 2.0 * pt * pph * tf_t)
        return d_r, d_t


def derivative(hole, y):
    """hamiltonian.rs:13-35.  y = (t, r, th, ph, pt, pr, pth, pph); returns the same layout."""
    r, th = y[1], y[2]
    gtt, gtr, gtf, grr, grf, ghh, gff = hole.inverse(r, th)
    pt, pr, pth, pph = y[4], y[5], y[6], y[7]
    # g[0] p0 + g[1] p1 + g[3] p3 ; g[4] p0 + g[5] p1 + g[7] p3 ; g[10] p2 ; g[12] p0 + g[13] p1 + g[15] p3
    dt = gtt * pt + gtr * pr + gtf * pph
    dr = gtr * pt + grr * pr + grf * pph
    dth = ghh * pth
    dph = gtf * pt + grf * pr + gff * pph
    d_r, d_t = hole.dH(r, th, (pt, pr, pth, pph))
    return (dt, dr, dth, dph, 0.0, -d_r, -d_t, 0.0)


def _axpy(y, terms):
    """state + sum_k k_k * s_k, in the reference's order: n[i] += k1[i]*s1 + k2[i]*s2 + ... (mod.rs:70-148)"""
    out = []
    for i in range(8):
        acc = terms[0][0][i] * terms[0][1]
        for k, s in terms[1:]:
            acc = acc + k[i] * s
        out.append(y[i] + acc)
    return tuple(out)


def fehlberg_try(hole, y, h):
    """integrator.rs:113-190 -> (5th-order state, max-abs error over the four coordinates)"""
    k1 = derivative(hole, y)
    k2 = derivative(hole, _axpy(y, [(k1, h / 4.0)]))
    k3 = derivative(hole, _axpy(y, [(k1, 3.0 * h / 32.0), (k2, 9.0 * h / 32.0)]))
    k4 = derivative(hole, _axpy(y, [(k1, 1932.0 * h / 2197.0), (k2, -7200.0 * h / 2197.0), (k3, 7296.0 * h / 2197.0)]))
    k5 = derivative(hole, _axpy(y, [(k1, 439.0 * h / 216.0), (k2, -8.0 * h), (k3, 3680.0 * h / 513.0),
                                    (k4, -845.0 * h / 4104.0)]))
    k6 = derivative(hole, _axpy(y, [(k1, -8.0 * h / 27.0), (k2, 2.0 * h), (k3, -3544.0 * h / 2565.0),
                                    (k4, 1859.0 * h / 4104.0), (k5, -11.0 * h / 40.0)]))
    new = []
    for i in range(8):
        new.append(y[i] + h * (16.0 / 135.0 * k1[i] + 6656.0 / 12825.0 * k3[i] + 28561.0 / 56430.0 * k4[i]
                               - 9.0 / 50.0 * k5[i] + 2.0 / 55.0 * k6[i]))
    err = 0.0
    for i in range(4):
        e = h * ((16.0 / 135.0 - 25.0 / 216.0) * k1[i] + (6656.0 / 12825.0 - 1408.0 / 2565.0) * k3[i]
                 + (28561.0 / 56430.0 - 2197.0 / 4104.0) * k4[i] + (-9.0 / 50.0 + 1.0 / 5.0) * k5[i]
                 + 2.0 / 55.0 * k6[i])
        err = rmax(err, abs(e))
    return tuple(new), err


def controlled_step(hole, y, h_try, tolerance, counters=None):
    """AdaptiveStepper::step, integrator.rs:69-108 (safety 0.9, min step 1e-5, max step 10) -> (state, next h)"""
    h = rclamp(h_try, -10.0, 10.0)
    while True:
        new, err = fehlberg_try(hole, y, h)
        if counters is not None:
            counters["tries"] += 1
        ratio = 0.0 if err == 0.0 else err / tolerance
        if ratio <= 1.0:
            growth = 5.0 if ratio < 1e-4 else 0.9 * rpow(ratio, -0.2)
            return new, rclamp(h * rmin(growth, 5.0), -10.0, 10.0)
        h = h * rmax(0.9 * rpow(ratio, -0.25), 0.1)
        if abs(h) < 1e-5:
            forced, _ = fehlberg_try(hole, y, 1e-5 * rsignum(h))
            if counters is not None:
                counters["tries"] += 1
                counters["forced"] += 1
            return forced, 1e-5 * rsignum(h)


def hamiltonian(hole, y):  # invariants/mod.rs:25-37
    gtt, gtr, gtf, grr, grf, ghh, gff = hole.inverse(y[1], y[2])
    pt, pr, pth, pph = y[4], y[5], y[6], y[7]
    return 0.5 * (gtt * pt * pt + grr * pr * pr + ghh * pth * pth + gff * pph * pph
                  + 2.0 * gtf * pt * pph + 2.0 * gtr * pt * pr + 2.0 * grf * pr * pph)


def renormalize(hole, y):  # invariants/renormalization.rs:13-45
    gtt, gtr, gtf, grr, grf, ghh, gff = hole.inverse(y[1], y[2])
    pt, pr, pth, pph = y[4], y[5], y[6], y[7]
    A = grr
    B = 2.0 * (gtr * pt + grf * pph)
    C = gtt * pt * pt + ghh * pth * pth + gff * pph * pph + 2.0 * gtf * pt * pph
    if abs(A) > 1e-12:
        disc = B * B - 4.0 * A * C
        if disc >= 0.0:
            sq = math.sqrt(disc)
            s1 = (-B + sq) / (2.0 * A)
            s2 = (-B - sq) / (2.0 * A)
            pr = s1 if abs(s1 - pr) < abs(s2 - pr) else s2
    return (y[0], y[1], y[2], y[3], pt, pr, pth, pph)


def integrate(y0, hole, tolerance=1e-8, initial_step=0.01, max_steps=10000, escape_radius=1000.0,
              renormalize_interval=10, counters=None):
    """integrate(), mod.rs:180-253 with IntegrationMethod::AdaptiveRKF45 -> (state, termination, steps, max drift)"""
    y = tuple(float(v) for v in y0)
    h = initial_step
    limit = hole.horizon() * 1.001  # check_termination, mod.rs:256-265
    drift = 0.0
    steps = 0
    y = renormalize(hole, y)
    for _ in range(max_steps):
        r = y[1]
        if r < limit:
            return y, HORIZON, steps, drift
        if r > escape_radius:
            return y, ESCAPE, steps, drift
        y, h = controlled_step(hole, y, h, tolerance, counters)
        if steps % renormalize_interval == 0:
            y = renormalize(hole, y)
        hv = abs(hamiltonian(hole, y))
        if hv > drift:
            drift = hv
        steps += 1
    return y, MAXSTEPS, steps, drift
