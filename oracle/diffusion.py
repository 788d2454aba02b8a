"""Oracle: schedules and DDIM samplers (test infrastructure, see oracle/__init__.py).

Restates mogen/models/utils/gaussian_diffusion.py for the inference configuration
(model_mean_type=START_X, model_var_type=FIXED_LARGE, eta=0, clip_denoised=False,
classifier_free_guidance_scale=0).  Tables are float64 numpy exactly as in the
reference and are cast to float32 at the point of use (`_extract_into_tensor`,
gaussian_diffusion.py:1613-1626).
"""
import numpy as np
import torch


def get_named_beta_schedule(name, n):
    """reference: gaussian_diffusion.py:229-268"""
    if name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "scaled_linear":
        return np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, n, dtype=np.float64) ** 2
    raise NotImplementedError(name)


def space_timesteps(num_timesteps, section_counts):
    """reference: gaussian_diffusion.py:1629-1711 (comma-separated section counts)."""
    section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx = 0
    all_steps = []
    for i, section_count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < section_count:
            raise ValueError("cannot divide section of %d steps into %d" % (size, section_count))
        frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
        cur_idx = 0.0
        taken = []
        for _ in range(section_count):
            taken.append(start_idx + round(cur_idx))
            cur_idx += frac_stride
        all_steps += taken
        start_idx += size
    return set(all_steps)


class SpacedSchedule:
    """reference: SpacedDiffusion.__init__ (gaussian_diffusion.py:1714-1738) over
    GaussianDiffusion.__init__ (:382-440)."""

    def __init__(self, beta_scheduler="scaled_linear", diffusion_steps=1000, respace="15,15,8,6,6"):
        base_betas = get_named_beta_schedule(beta_scheduler, diffusion_steps)
        base_ac = np.cumprod(1.0 - base_betas, axis=0)
        use = space_timesteps(diffusion_steps, respace)
        last = 1.0
        new_betas = []
        self.timestep_map = []
        for i, ac in enumerate(base_ac):
            if i in use:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        betas = np.array(new_betas, dtype=np.float64)
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        # q(x_{t-1} | x_t, x_0) (gaussian_diffusion.py:427-440)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(1.0 - betas) / (1.0 - self.alphas_cumprod)


def _ext(arr, i):
    """_extract_into_tensor for a batch that shares one step index: fp64 -> fp32 scalar tensor."""
    return torch.tensor(arr[i], dtype=torch.float64).float()


def q_sample(sch, x_start, i, noise):
    """reference: gaussian_diffusion.py:459-477"""
    return _ext(sch.sqrt_alphas_cumprod, i) * x_start + _ext(sch.sqrt_one_minus_alphas_cumprod, i) * noise


def eps_from_xstart(sch, x, i, x0):
    """reference: gaussian_diffusion.py:693-697"""
    return (_ext(sch.sqrt_recip_alphas_cumprod, i) * x - x0) / _ext(sch.sqrt_recipm1_alphas_cumprod, i)


def ddim_sample(sch, model, x, i, noise, in_seq=None):
    """reference: gaussian_diffusion.py:910-1001 with eta=0.  `model(x, t_orig)` returns the
    x0 prediction; `noise(shape)` draws in the reference's order (randn_like(in_seq) if
    in_seq is given, then randn_like(x))."""
    if in_seq is not None:
        nz = (in_seq != 0).any(dim=-1)
        x = x * (~nz).to(torch.int).unsqueeze(-1).float()
        x_t = q_sample(sch, in_seq, i, noise(in_seq.shape))
        x = x + x_t * nz.to(torch.int).unsqueeze(-1).float()
    t_orig = torch.full((x.shape[0],), sch.timestep_map[i], dtype=torch.long)
    x0 = model(x, t_orig)
    eps = eps_from_xstart(sch, x, i, x0)
    ab = _ext(sch.alphas_cumprod, i)
    ab_prev = _ext(sch.alphas_cumprod_prev, i)
    sigma = 0.0 * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
    n = noise(x.shape)
    mean_pred = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    nonzero = 0.0 if i == 0 else 1.0
    return mean_pred + nonzero * sigma * n, x0


def p_sample(sch, model, x, i, noise):
    """reference: gaussian_diffusion.py:741-803 `p_sample` over :503-653 `p_mean_variance` with
    model_mean_type START_X (pred_xstart = model output, mean = q_posterior_mean, :479-501) and
    model_var_type FIXED_LARGE (log variance = log(append(posterior_variance[1], betas[1:])), :560-570)."""
    t_orig = torch.full((x.shape[0],), sch.timestep_map[i], dtype=torch.long)
    x0 = model(x, t_orig)
    mean = _ext(sch.posterior_mean_coef1, i) * x0 + _ext(sch.posterior_mean_coef2, i) * x
    log_var = _ext(np.log(np.append(sch.posterior_variance[1], sch.betas[1:])), i)
    n = noise(x.shape)
    nonzero = 0.0 if i == 0 else 1.0
    return mean + nonzero * torch.exp(0.5 * log_var) * n, x0


def p_sample_loop(sch, model, img, noise):
    """reference: gaussian_diffusion.py:805-905 (img = th.randn(*shape) already drawn)."""
    for i in range(sch.num_timesteps - 1, -1, -1):
        img, _ = p_sample(sch, model, img, i, noise)
    return img


def ddim_reverse_sample(sch, model, x, i):
    """reference: gaussian_diffusion.py:1003-1040"""
    t_orig = torch.full((x.shape[0],), sch.timestep_map[i], dtype=torch.long)
    x0 = model(x, t_orig)
    eps = eps_from_xstart(sch, x, i, x0)
    ab_next = _ext(sch.alphas_cumprod_next, i)
    return x0 * torch.sqrt(ab_next) + torch.sqrt(1 - ab_next) * eps


def ddim_sample_loop(sch, model, img, noise, in_seq=None, trace=None):
    """reference: gaussian_diffusion.py:1042-1135 (img = start noise already drawn)."""
    for i in range(sch.num_timesteps - 1, -1, -1):
        img, x0 = ddim_sample(sch, model, img, i, noise, in_seq)
        if trace is not None:
            trace.append(img)
    return img


def ddim_reverse_sample_loop(sch, model, start_img):
    """reference: gaussian_diffusion.py:1137-1230 with return_all_timesteps=True:
    list index k holds the latent at level alphas_cumprod_next[k]."""
    img = start_img
    out = []
    for i in range(sch.num_timesteps):
        img = ddim_reverse_sample(sch, model, img, i)
        out.append(img)
    return out


def retrieval_guidance_update(img, in_seq, g_iter, lr):
    """reference: gaussian_diffusion.py:1263-1273, 1351-1378.  The reference differentiates
    mse_loss(x*mask, in_seq) with autograd g_iter times; the gradient of that mean-squared
    error is 2*mask*(mask*x - in_seq)/numel, restated here without autograd."""
    mask = (in_seq != 0).any(dim=-1).unsqueeze(-1).float()
    x = img.clone()
    numel = float(x.numel())
    for _ in range(g_iter):
        grad = (2.0 / numel) * ((x * mask - in_seq) * mask)
        x = x - lr * grad
    return x


def ddim_guided_sample_loop(sch, model, img, noise, guidance_iters, inverted_latent_list,
                            guidance_lr, in_seq=None, trace=None):
    """reference: gaussian_diffusion.py:1233-1395.  Note the reference overwrites its local
    `in_seq` with inverted_latent_list[i] on every step but the first, so ddim_sample
    re-inserts q_sample(inverted_latent_list[i]) on the masked rows."""
    assert len(guidance_iters) == len(inverted_latent_list)
    first = sch.num_timesteps - 1
    for i in range(first, -1, -1):
        if i != first:
            in_seq = inverted_latent_list[i]
            img = retrieval_guidance_update(img, in_seq, guidance_iters[i], guidance_lr)
        img, x0 = ddim_sample(sch, model, img, i, noise, in_seq)
        if trace is not None:
            trace.append(img)
    return img
