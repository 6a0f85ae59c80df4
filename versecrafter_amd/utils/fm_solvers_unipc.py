"""FlowUniPCMultistepScheduler: UniPC (B(h) = e^h - 1, "bh2") multistep predictor-corrector for
flow-matching models, as the reference's sampler uses it
(pipeline_wan_versecrafter.py:32, 750-752, 909; inference/versecrafter_inference.py:252-261).

The reference imports this class from the un-vendored videox_fun.utils.fm_solvers_unipc (origin: Wan2.1
wan/utils/fm_solvers_unipc.py); its source is not in the reference tree, so this is a restatement of the
published algorithm from the call-site contract -- "parity unpinned" (SURVEY.md Appendix C).  The update is
elementwise on the [1,16,T,h,w] latent: torch ops on whatever device the latent lives on.

Contract used by the pipeline:  set_timesteps(n, device=, shift=) -> .timesteps (int64 [n]);
step(model_output, t, sample, return_dict=False)[0] -> next sample in sample.dtype;  .order == 1.
"""
from typing import List, Optional, Tuple, Union

import numpy as np
import torch


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: Optional[float] = 1.0, use_dynamic_shifting: bool = False, predict_x0: bool = True,
                 solver_type: str = "bh2", lower_order_final: bool = True, disable_corrector: List[int] = [],
                 final_sigmas_type: str = "zero", **unused):
        if solver_type not in ("bh1", "bh2"):
            raise NotImplementedError(f"{solver_type} is not implemented")
        if prediction_type != "flow_prediction" or not predict_x0:
            raise NotImplementedError("only flow_prediction with predict_x0=True (the Wan2.1 configuration)")
        self.config = type("Cfg", (), dict(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                                           prediction_type=prediction_type, shift=shift,
                                           use_dynamic_shifting=use_dynamic_shifting, predict_x0=predict_x0,
                                           solver_type=solver_type, lower_order_final=lower_order_final,
                                           final_sigmas_type=final_sigmas_type))()
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = 1.0 - alphas
        if not use_dynamic_shifting:
            sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.sigmas = torch.from_numpy(sigmas.astype(np.float32))
        self.timesteps = self.sigmas * num_train_timesteps
        self.sigma_min = float(self.sigmas[-1])
        self.sigma_max = float(self.sigmas[0])
        self.num_inference_steps = None
        self.model_outputs = [None] * solver_order
        self.timestep_list = [None] * solver_order
        self.lower_order_nums = 0
        self.disable_corrector = list(disable_corrector)
        self.last_sample = None
        self._step_index = None
        self.this_order = 1

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas=None, mu=None,
                      shift: Optional[float] = None):
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if self.config.use_dynamic_shifting:
            sigmas = np.exp(mu) / (np.exp(mu) + (1 / sigmas - 1))
        else:
            if shift is None:
                shift = self.config.shift
            sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        sigma_last = 0.0 if self.config.final_sigmas_type == "zero" else float(sigmas[-1])
        timesteps = sigmas * self.config.num_train_timesteps
        sigmas = np.concatenate([sigmas, [sigma_last]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)                       # kept on the host, as upstream
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.config.solver_order
        self.timestep_list = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = None

    def scale_model_input(self, sample: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        return sample

    # -- internals ---------------------------------------------------------------------------------
    def _init_step_index(self, timestep):
        ts = self.timesteps
        t = timestep.to(ts.device) if torch.is_tensor(timestep) else torch.tensor(timestep, device=ts.device)
        idx = (ts == t).nonzero()
        if len(idx) == 0:
            self._step_index = len(ts) - 1
        else:
            self._step_index = int(idx[1 if len(idx) > 1 else 0])

    @staticmethod
    def _lambda(sigma: torch.Tensor) -> torch.Tensor:
        return torch.log(1 - sigma) - torch.log(sigma)              # alpha_t = 1 - sigma, sigma_t = sigma

    def _coeffs(self, order: int, rks: torch.Tensor, hh: torch.Tensor):
        """R (Vandermonde of rks) and b of the UniPC linear system for B(h) of the configured type."""
        h_phi_1 = torch.expm1(hh)
        B_h = hh if self.config.solver_type == "bh1" else torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return torch.stack(R), torch.stack(b), h_phi_1, B_h

    def _predict(self, sample: torch.Tensor, order: int) -> torch.Tensor:
        m0 = self.model_outputs[-1]
        x = sample
        sigma_t, sigma_s0 = self.sigmas[self._step_index + 1], self.sigmas[self._step_index]
        alpha_t = 1 - sigma_t
        h = self._lambda(sigma_t) - self._lambda(sigma_s0)
        rks, D1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lambda(self.sigmas[self._step_index - i]) - self._lambda(sigma_s0)) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in rks])
        hh = -h
        R, b, h_phi_1, B_h = self._coeffs(order, rks, hh)
        x_t_ = (sigma_t / sigma_s0) * x - alpha_t * h_phi_1 * m0
        if D1s:
            rhos_p = torch.tensor([0.5], dtype=x.dtype) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1]).to(x.dtype)
            pred_res = sum(float(r) * D for r, D in zip(rhos_p, D1s))
        else:
            pred_res = 0
        x_t = x_t_ - alpha_t * B_h * pred_res
        return x_t.to(x.dtype)

    def _correct(self, this_model_output: torch.Tensor, last_sample: torch.Tensor, this_sample: torch.Tensor,
                 order: int) -> torch.Tensor:
        m0 = self.model_outputs[-1]
        x, model_t = last_sample, this_model_output
        sigma_t, sigma_s0 = self.sigmas[self._step_index], self.sigmas[self._step_index - 1]
        alpha_t = 1 - sigma_t
        h = self._lambda(sigma_t) - self._lambda(sigma_s0)
        rks, D1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lambda(self.sigmas[self._step_index - (i + 1)]) - self._lambda(sigma_s0)) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in rks])
        hh = -h
        R, b, h_phi_1, B_h = self._coeffs(order, rks, hh)
        rhos_c = torch.tensor([0.5], dtype=x.dtype) if order == 1 else torch.linalg.solve(R, b).to(x.dtype)
        x_t_ = (sigma_t / sigma_s0) * x - alpha_t * h_phi_1 * m0
        corr_res = sum(float(r) * D for r, D in zip(rhos_c[:-1], D1s)) if D1s else 0
        D1_t = model_t - m0
        x_t = x_t_ - alpha_t * B_h * (corr_res + float(rhos_c[-1]) * D1_t)
        return x_t.to(x.dtype)

    # -- public ------------------------------------------------------------------------------------
    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True,
             generator=None) -> Union[Tuple[torch.Tensor], dict]:
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        if self._step_index is None:
            self._init_step_index(timestep)
        use_corrector = (self._step_index > 0 and self._step_index - 1 not in self.disable_corrector
                         and self.last_sample is not None)
        # flow prediction -> x0 (uses the un-corrected sample, as upstream)
        sigma_t = self.sigmas[self._step_index]
        x0_pred = sample - sigma_t * model_output
        if use_corrector:
            sample = self._correct(x0_pred, self.last_sample, sample, self.this_order)
        for i in range(self.config.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
            self.timestep_list[i] = self.timestep_list[i + 1]
        self.model_outputs[-1] = x0_pred
        self.timestep_list[-1] = timestep
        if self.config.lower_order_final:
            this_order = min(self.config.solver_order, len(self.timesteps) - self._step_index)
        else:
            this_order = self.config.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        assert self.this_order > 0
        self.last_sample = sample
        prev_sample = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        if not return_dict:
            return (prev_sample,)
        return {"prev_sample": prev_sample}

    # -- fused device path (libvcengine: vc_op_unipc_update) --------------------------------------------
    def _corr_scalars(self, order: int, dtype):
        """The scalars `_correct` multiplies tensors with, computed by the same 0-dim fp32 tensor arithmetic."""
        sigma_t, sigma_s0 = self.sigmas[self._step_index], self.sigmas[self._step_index - 1]
        alpha_t = 1 - sigma_t
        h = self._lambda(sigma_t) - self._lambda(sigma_s0)
        rks = []
        for i in range(1, order):
            rks.append((self._lambda(self.sigmas[self._step_index - (i + 1)]) - self._lambda(sigma_s0)) / h)
        rk = float(rks[0]) if rks else 1.0
        rks.append(torch.tensor(1.0))
        rks = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in rks])
        R, b, h_phi_1, B_h = self._coeffs(order, rks, -h)
        rhos_c = torch.tensor([0.5], dtype=dtype) if order == 1 else torch.linalg.solve(R, b).to(dtype)
        return (float(sigma_t / sigma_s0), float(alpha_t * h_phi_1), float(alpha_t * B_h), rk, float(rhos_c[0]),
                float(rhos_c[-1]))

    def _pred_scalars(self, order: int):
        sigma_t, sigma_s0 = self.sigmas[self._step_index + 1], self.sigmas[self._step_index]
        alpha_t = 1 - sigma_t
        h = self._lambda(sigma_t) - self._lambda(sigma_s0)
        rk = 1.0
        if order == 2:
            rk = float((self._lambda(self.sigmas[self._step_index - 1]) - self._lambda(sigma_s0)) / h)
        elif order > 2:
            raise NotImplementedError("fused UniPC update: solver_order <= 2 (the Wan configuration)")
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = hh if self.config.solver_type == "bh1" else torch.expm1(hh)
        return float(sigma_t / sigma_s0), float(alpha_t * h_phi_1), float(alpha_t * B_h), rk, 0.5

    def step_cfg(self, noise_pred: torch.Tensor, timestep, sample: torch.Tensor, guidance_scale: Optional[float] = None):
        """`step` preceded by the classifier-free-guidance combine of PIPE.py:903-906, as ONE HIP kernel over the latent
        (bf16 CUDA tensors only).  noise_pred: [2B, ...] = [uncond, cond] when guidance_scale is given, else [B, ...].
        Bit-identical to `cfg combine; step()`: the kernel rounds to bf16 wherever the torch ops do.  Returns the next sample."""
        from .. import ops
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' first")
        if not (sample.is_cuda and sample.dtype == torch.bfloat16 and noise_pred.dtype == torch.bfloat16):
            raise TypeError("step_cfg runs on bfloat16 CUDA (HIP) tensors; use step() otherwise")
        if self.config.solver_order > 2:
            raise NotImplementedError("fused UniPC update: solver_order <= 2 (the Wan configuration)")
        if self._step_index is None:
            self._init_step_index(timestep)
        i = self._step_index
        use_corrector = i > 0 and i - 1 not in self.disable_corrector and self.last_sample is not None
        corr_order = self.this_order
        flags = (1 if guidance_scale is not None else 0) | (2 if use_corrector else 0)
        sc = [0.0 if guidance_scale is None else float(guidance_scale), float(self.sigmas[i])]
        if use_corrector:
            sc += list(self._corr_scalars(corr_order, sample.dtype))
            flags |= 4 if corr_order == 2 else 0
        else:
            sc += [0.0, 0.0, 0.0, 1.0, 0.0, 0.0]
        if self.config.lower_order_final:
            this_order = min(self.config.solver_order, len(self.timesteps) - i)
        else:
            this_order = self.config.solver_order
        this_order = min(this_order, self.lower_order_nums + 1)
        sc += list(self._pred_scalars(this_order))
        flags |= 8 if this_order == 2 else 0
        m_old, m_older = self.model_outputs[-1], self.model_outputs[-2]
        x0, corrected, nxt = ops.unipc_update(noise_pred, sample, sc, flags, last=self.last_sample, m0=m_old, m1=m_older)
        for k in range(self.config.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
            self.timestep_list[k] = self.timestep_list[k + 1]
        self.model_outputs[-1] = x0
        self.timestep_list[-1] = timestep
        self.this_order = this_order
        self.last_sample = corrected if use_corrector else sample
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return nxt

    def __len__(self):
        return self.config.num_train_timesteps
