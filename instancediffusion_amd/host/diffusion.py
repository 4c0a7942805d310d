"""Host mirror of ``ldm/models/diffusion/ddpm.py`` + ``ldm.py``: the noise-schedule buffers the samplers read
(``betas``, ``alphas_cumprod``, ``alphas_cumprod_prev``, ``num_timesteps``) and ``q_sample``.  Tiny, host-side."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """util.py:30-52 (float64 numpy)."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "cosine":
        ts = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        al = torch.cos(ts / (1 + cosine_s) * np.pi / 2).pow(2)
        al = al / al[0]
        betas = torch.from_numpy(np.clip((1 - al[1:] / al[:-1]).numpy(), a_min=0, a_max=0.999))
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=False):
    """util.py:55-69: 'uniform' -> range(0, T, T // S) + 1."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    return steps + 1


class DDPM(nn.Module):
    """ddpm.py:11-54."""

    def __init__(self, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
        super().__init__()
        self.v_posterior = 0
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end,
                                   cosine_s=cosine_s)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end

        def reg(name, v):
            self.register_buffer(name, torch.tensor(v, dtype=torch.float32))
        reg("betas", betas)
        reg("alphas_cumprod", ac)
        reg("alphas_cumprod_prev", ac_prev)
        reg("sqrt_alphas_cumprod", np.sqrt(ac))
        reg("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - ac))
        reg("log_one_minus_alphas_cumprod", np.log(1.0 - ac))
        reg("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / ac))
        reg("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / ac - 1))
        pv = (1 - self.v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + self.v_posterior * betas
        reg("posterior_variance", pv)
        reg("posterior_log_variance_clipped", np.log(np.maximum(pv, 1e-20)))
        reg("posterior_mean_coef1", betas * np.sqrt(ac_prev) / (1.0 - ac))
        reg("posterior_mean_coef2", (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))


class LatentDiffusion(DDPM):
    """ldm.py:11-20."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.clip_denoised = False

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        shape = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
        a = self.sqrt_alphas_cumprod.gather(-1, t).reshape(shape)
        b = self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(shape)
        return a * x_start + b * noise
