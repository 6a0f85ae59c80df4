"""CPU ORACLE (test infrastructure) for the sampler arithmetic around the hot path: the flow-matching
UniPC-bh2 update (third-party videox_fun.utils.fm_solvers_unipc.FlowUniPCMultistepScheduler; source absent
from the reference tree -> restated from the published UniPC algorithm, "parity unpinned") and the
classifier-free-guidance combine of pipeline_wan_versecrafter.py:904-906.  float64 numpy, closed forms for
the orders the Wan configuration uses (solver_order=2, lower_order_final)."""
import numpy as np


def flow_sigmas(num_steps: int, shift: float, num_train_timesteps: int = 1000):
    """set_timesteps(n, shift=s): sigma = linspace(sigma_max, sigma_min, n+1)[:-1] -> shift*s/(1+(shift-1)s), final 0.
    Returns (sigmas[n+1] float32-rounded like upstream, timesteps[n] int64)."""
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1]
    base = (1.0 - alphas).astype(np.float32)              # constructor with shift = 1 (CLI.py:257-261)
    smax, smin = float(base[0]), float(base[-1])
    s = np.linspace(smax, smin, num_steps + 1)[:-1]
    s = shift * s / (1 + (shift - 1) * s)
    ts = (s * num_train_timesteps).astype(np.int64)
    return np.concatenate([s, [0.0]]).astype(np.float32).astype(np.float64), ts


def _lam(s):
    with np.errstate(divide="ignore"):
        return np.log(1 - s) - np.log(s)


class UniPCOracle:
    def __init__(self, num_steps: int, shift: float):
        self.sigmas, self.timesteps = flow_sigmas(num_steps, shift)
        self.n = num_steps
        self.i = 0
        self.m = []            # x0 predictions, newest last
        self.last_sample = None
        self.order_used = 1
        self.lower = 0

    def _coeffs(self, order, rks, hh):
        with np.errstate(invalid="ignore", divide="ignore"):
            h_phi_1 = np.expm1(hh)
            B_h = np.expm1(hh)
            h_phi_k = h_phi_1 / hh - 1
            R, b, fact = [], [], 1
            for k in range(1, order + 1):
                R.append(np.power(rks, k - 1))
                b.append(h_phi_k * fact / B_h)
                fact *= k + 1
                h_phi_k = h_phi_k / hh - 1 / fact
        return np.stack(R), np.array(b), h_phi_1, B_h

    def step(self, v: np.ndarray, x: np.ndarray) -> np.ndarray:
        i, s = self.i, self.sigmas
        x0 = x - s[i] * v
        if i > 0 and self.last_sample is not None:            # corrector (UniC) with the order of the last predictor
            order = self.order_used
            st, ss = s[i], s[i - 1]
            h = _lam(st) - _lam(ss)
            m0 = self.m[-1]
            rks, D1s = [], []
            for k in range(1, order):
                rk = (_lam(s[i - (k + 1)]) - _lam(ss)) / h
                rks.append(rk)
                D1s.append((self.m[-(k + 1)] - m0) / rk)
            rks.append(1.0)
            R, b, h_phi_1, B_h = self._coeffs(order, np.array(rks), -h)
            rho = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
            corr = sum(r * D for r, D in zip(rho[:-1], D1s)) if D1s else 0.0
            x = (st / ss) * self.last_sample - (1 - st) * h_phi_1 * m0 - (1 - st) * B_h * (corr + rho[-1] * (x0 - m0))
        self.m.append(x0)
        self.m = self.m[-2:]
        order = min(2, self.n - i)                              # lower_order_final
        order = min(order, self.lower + 1)                      # multistep warm-up
        self.order_used = order
        self.last_sample = x
        st, ss = s[i + 1], s[i]
        h = _lam(st) - _lam(ss)
        m0 = self.m[-1]
        with np.errstate(invalid="ignore"):
            h_phi_1 = np.expm1(-h)
            B_h = np.expm1(-h)
            x_next = (st / ss) * x - (1 - st) * h_phi_1 * m0
            if order == 2:
                rk = (_lam(s[i - 1]) - _lam(ss)) / h
                x_next = x_next - (1 - st) * B_h * 0.5 * (self.m[-2] - m0) / rk
        if self.lower < 2:
            self.lower += 1
        self.i += 1
        return x_next


def cfg_combine(uncond: np.ndarray, cond: np.ndarray, g: float) -> np.ndarray:
    """pipeline_wan_versecrafter.py:904-906."""
    return uncond + g * (cond - uncond)
