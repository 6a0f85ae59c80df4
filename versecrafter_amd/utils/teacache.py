"""TeaCache step-skipping gate.

Host-side restatement of videox_fun.models.cache_utils.TeaCache (third-party, un-vendored: parity
unpinned) with the decision logic of the reference's _process_teacache_skip_logic
(versecrafter/models/wan_transformer3d.py:205-245) folded into `gate`.  The residual tensor itself
(previous_residual_cond, VC.py:390-411) is kept inside the HIP engine.
"""
import numpy as np
import torch


class TeaCache:
    def __init__(self, coefficients, num_steps: int, rel_l1_thresh: float = 0.0, num_skip_start_steps: int = 0,
                 offload: bool = True):
        if num_steps < 1:
            raise ValueError(f"`num_steps` must be greater than 0 but is {num_steps}.")
        if rel_l1_thresh < 0:
            raise ValueError(f"`rel_l1_thresh` must be greater than or equal to 0 but is {rel_l1_thresh}.")
        if num_skip_start_steps < 0 or num_skip_start_steps > num_steps:
            raise ValueError("`num_skip_start_steps` must be in [0, num_steps]")
        self.coefficients = coefficients
        self.num_steps = num_steps
        self.rel_l1_thresh = rel_l1_thresh
        self.num_skip_start_steps = num_skip_start_steps
        self.offload = offload          # accepted for API compatibility; the residual never leaves HBM here
        self.rescale_func = np.poly1d(coefficients)
        self.reset()

    @staticmethod
    def compute_rel_l1_distance(prev: torch.Tensor, cur: torch.Tensor) -> float:
        return ((cur - prev).abs().mean() / prev.abs().mean()).cpu().item()

    def reset(self):
        self.cnt = 0
        self.should_calc = True
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None

    def gate(self, modulated_inp: torch.Tensor) -> bool:
        """WT.py:219-243 with cond_flag=True: returns should_calc for this step."""
        if self.cnt < self.num_skip_start_steps:
            should_calc = True
            self.accumulated_rel_l1_distance = 0
        else:
            rel = self.compute_rel_l1_distance(self.previous_modulated_input, modulated_inp)
            self.accumulated_rel_l1_distance += self.rescale_func(rel)
            if self.accumulated_rel_l1_distance < self.rel_l1_thresh:
                should_calc = False
            else:
                should_calc = True
                self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = modulated_inp
        self.should_calc = should_calc
        return should_calc
