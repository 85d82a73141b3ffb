"""Oracle (test infrastructure): CPU restatement of the diffusion objective.

Follows /root/reference/tools/gaussian_diffusion.py:
  * beta schedules            :59-123   (get_named_beta_schedule, betas_for_alpha_bar)
  * coefficient tables        :168-205  (GaussianDiffusion.__init__)
  * table gather              :1059-1072 (_extract_into_tensor: float64 gather, THEN .float())
  * q_sample                  :234-252
  * sample_t                  :810-816
  * compute_target            :818-832
  * training_losses (MSE, learned-variance vb term, KL)  :834-930
  * q_posterior_mean_variance :254-276, p_mean_variance (training side) :278-384, _predict_xstart_* :386-410
  * _vb_terms_bpd             :775-808
  * sampling: p_mean_variance :278-384, p_sample(+loops) :461-601, ddim_sample(+loops) :603-790 (no cond_fn)
and tools/losses.py:12-76 (normal_kl, approx_standard_normal_cdf, discretized_gaussian_log_likelihood)
  * compute_mse_loss_weight   :1092-1148
  * FlowMatching training     :1151-1340
and tools/nn.py:86-90 (mean_flat).  Pinned by tests/golden/diffusion_*.npz, objective.pt, vb_objective.pt.
"""
import enum
import math

import numpy as np
import torch


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()
    VELOCITY = enum.auto()
    VECTOR = enum.auto()
    SCORE = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


def betas_for_alpha_bar(T, alpha_bar, max_beta=0.999):
    out = np.empty(T, dtype=np.float64)
    for i in range(T):
        lo, hi = i / T, (i + 1) / T
        out[i] = min(1 - alpha_bar(hi) / alpha_bar(lo), max_beta)
    return out


def get_named_beta_schedule(name, T, lambda_max=10.0, lambda_min=-10.0):
    if name == "linear":
        s = 1000 / T
        return np.linspace(s * 0.0001, s * 0.02, T, dtype=np.float64)
    if name == "cosine":
        return betas_for_alpha_bar(T, lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2)
    if name == "linear_logsnr":
        def abar(u):
            lam = lambda_max + u * (lambda_min - lambda_max)
            return 1.0 / (1.0 + math.exp(-lam))
        return betas_for_alpha_bar(T, abar)
    raise NotImplementedError(f"unknown beta schedule: {name}")


def mean_flat(x):
    return x.mean(dim=list(range(1, x.dim())))


def extract(arr, t, shape):
    """float64 table -> gather at t -> fp32 -> broadcast view of `shape`."""
    v = torch.from_numpy(arr).to(t.device)[t].float()
    while v.dim() < len(shape):
        v = v[..., None]
    return v.expand(shape)


def _parse_k(wt, prefix):
    return float(wt.split(prefix)[-1])


def compute_mse_loss_weight(mean_type, wt, t, alpha, sigma, p2_k=1.0, p2_gamma=1.0):
    """reference :1092-1148.  NOTE the aliasing it has: for EPSILON/'lambda' the
    returned tensor IS `sigma`, and the snr==0 patch writes into it."""
    snr = (alpha / sigma) ** 2
    if wt == "constant":
        return torch.ones_like(t)
    name = mean_type.name
    w = None

    def with_k(k, op):
        pair = torch.stack([snr, k * torch.ones_like(t)], dim=1)
        return (pair.min(dim=1)[0] if op == "min" else pair.max(dim=1)[0])

    if name == "EPSILON":
        if wt.startswith("min_snr_"):
            w = with_k(_parse_k(wt, "min_snr_"), "min") / snr
        elif wt.startswith("max_snr_"):
            w = with_k(_parse_k(wt, "max_snr_"), "max") / snr
        elif wt == "lambda":
            w = sigma
        elif wt == "debias":
            w = sigma / alpha
        elif wt == "p2":
            w = 1 / (p2_k + snr) ** p2_gamma
        elif wt == "min_debias":
            w = torch.minimum(sigma / alpha, torch.ones_like(sigma))
        elif wt == "max_debias":
            w = torch.maximum(sigma / alpha, torch.ones_like(sigma))
    elif name == "START_X":
        if wt == "trunc_snr":
            w = torch.stack([snr, torch.ones_like(t)], dim=1).max(dim=1)[0]
        elif wt == "snr":
            w = snr
        elif wt == "inv_snr":
            w = 1.0 / snr
        elif wt.startswith("min_snr_"):
            w = with_k(_parse_k(wt, "min_snr_"), "min")
        elif wt.startswith("max_snr_"):
            w = with_k(_parse_k(wt, "max_snr_"), "max")
        elif wt == "lambda":
            w = alpha
    elif name == "VECTOR":
        if wt == "lambda":
            w = torch.ones_like(t)
    elif name == "VELOCITY":
        if wt.startswith("min_snr_"):
            w = with_k(_parse_k(wt, "min_snr_"), "min") / (snr + 1)
        elif wt == "lambda":
            w = alpha * sigma
    if w is None:
        raise ValueError(f"Invalid mse_loss_weight_type: {wt}")
    w[snr == 0] = 1.0
    return w


def normal_kl(mean1, logvar1, mean2, logvar2):
    """tools/losses.py:12-39"""
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def approx_standard_normal_cdf(x):
    """tools/losses.py:42-47"""
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * torch.pow(x, 3))))


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    """tools/losses.py:50-76: bins of width 2/255 around x in [-1, 1], open-ended beyond +-0.999."""
    assert x.shape == means.shape == log_scales.shape
    centered_x = x - means
    inv_stdv = torch.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv_stdv * (centered_x + 1.0 / 255.0))
    cdf_min = approx_standard_normal_cdf(inv_stdv * (centered_x - 1.0 / 255.0))
    log_cdf_plus = torch.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = torch.log((1.0 - cdf_min).clamp(min=1e-12))
    cdf_delta = cdf_plus - cdf_min
    return torch.where(x < -0.999, log_cdf_plus,
                       torch.where(x > 0.999, log_one_minus_cdf_min, torch.log(cdf_delta.clamp(min=1e-12))))


class GaussianDiffusion:
    def __init__(self, *, args, betas, model_mean_type, model_var_type, loss_type,
                 rescale_timesteps=False, device="cpu"):
        self.args = args
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        self.mse_loss_weight_type = args.weight_type
        self.gamma = args.gamma
        self.learn_sigma = args.learn_sigma
        self.p2_gamma = args.p2_gamma
        self.p2_k = args.p2_k

        b = np.array(betas, dtype=np.float64)
        assert b.ndim == 1 and (b >= 0).all() and (b <= 1).all()
        self.betas = b
        self.num_timesteps = int(b.shape[0])
        self.alphas = 1.0 - b
        ac = np.cumprod(self.alphas, axis=0)
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod_next = np.append(ac[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        self.posterior_variance = b * (1.0 - self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = b * np.sqrt(self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(self.alphas) / (1.0 - ac)

    def _scale_timesteps(self, t):
        return t.float() * (1000.0 / self.num_timesteps) if self.rescale_timesteps else t

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        return (extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def sample_t(self, x_start):
        if self.args.time_dist[0] == "uniform":
            return torch.randint(0, self.num_timesteps, (x_start.shape[0],), device=x_start.device)
        raise NotImplementedError(f"Unknown time_dist: {self.args.time_dist}")

    def compute_target(self, x_start, noise, t, alpha=None, sigma=None):
        if alpha is None or sigma is None:
            alpha = extract(self.sqrt_alphas_cumprod, t, t.shape)
            sigma = extract(self.sqrt_one_minus_alphas_cumprod, t, t.shape)
        mt = self.model_mean_type
        if mt == ModelMeanType.START_X:
            return x_start
        if mt == ModelMeanType.EPSILON:
            return noise
        if mt == ModelMeanType.VELOCITY:
            return alpha[:, None, None, None] * noise - sigma[:, None, None, None] * x_start
        if mt == ModelMeanType.PREVIOUS_X:
            x_t = self.q_sample(x_start, t, noise=noise)
            return (extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                    + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        raise KeyError(mt)

    def training_losses(self, model, x_start, features=None, t=None, model_kwargs=None, noise=None):
        model_kwargs = model_kwargs or {}
        if noise is None:
            noise = torch.randn_like(x_start)      # RNG draw #1 (reference :849)
        if t is None:
            t = self.sample_t(x_start)             # RNG draw #2 (reference :851)
        x_t = self.q_sample(x_start, t, noise=noise)
        alpha = extract(self.sqrt_alphas_cumprod, t, t.shape)
        sigma = extract(self.sqrt_one_minus_alphas_cumprod, t, t.shape)
        w = compute_mse_loss_weight(self.model_mean_type, self.mse_loss_weight_type, t, alpha, sigma,
                                    self.p2_k, self.p2_gamma)
        if self.loss_type in (LossType.KL, LossType.RESCALED_KL):                    # reference :865-876
            raw = model(x_t, self._scale_timesteps(t), **model_kwargs)
            loss = self._vb_terms_bpd(raw[0] if isinstance(raw, tuple) else raw, x_start, x_t, t)
            if self.loss_type == LossType.RESCALED_KL:
                loss = loss * self.num_timesteps
            return {"loss": loss}
        if self.loss_type not in (LossType.MSE, LossType.RESCALED_MSE):
            raise NotImplementedError(self.loss_type)
        raw = model(x_t, self._scale_timesteps(t), **model_kwargs)
        out = raw[0] if isinstance(raw, tuple) else raw
        terms = {}
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):   # reference :887-906
            B, C = x_t.shape[:2]
            assert out.shape == (B, C * 2, *x_t.shape[2:])
            out, var_values = torch.split(out, C, dim=1)
            # the bound trains the variance only: the mean prediction enters it detached
            terms["vb"] = self._vb_terms_bpd(torch.cat([out.detach(), var_values], dim=1), x_start, x_t, t)
            if self.loss_type == LossType.RESCALED_MSE:
                terms["vb"] = terms["vb"] * (self.num_timesteps / 1000.0)
        target = self.compute_target(x_start, noise, t, alpha, sigma)
        assert out.shape == target.shape == x_start.shape
        terms["mse"] = w * mean_flat((target - out) ** 2)
        terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
        return terms

    # ---- variational bound (reference :254-276, :278-384, :775-808) -----------------------------------------
    def q_posterior_mean_variance(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        mean = (extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, extract(self.posterior_variance, t, x_t.shape), extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def model_mean_log_variance(self, model_output, x, t):
        """The training-side half of p_mean_variance (no clipping, no denoised_fn): (mean, log variance) of p(x_{t-1}|x_t)."""
        B, C = x.shape[:2]
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            assert model_output.shape == (B, C * 2, *x.shape[2:])
            model_output, var_values = torch.split(model_output, C, dim=1)
            if self.model_var_type == ModelVarType.LEARNED:
                log_var = var_values
            else:
                min_log = extract(self.posterior_log_variance_clipped, t, x.shape)
                max_log = extract(np.log(self.betas), t, x.shape)
                frac = (var_values + 1) / 2                    # [-1, 1] -> [min_var, max_var]
                log_var = frac * max_log + (1 - frac) * min_log
        else:
            tab = {ModelVarType.FIXED_LARGE: np.log(np.append(self.posterior_variance[1], self.betas[1:])),
                   ModelVarType.FIXED_SMALL: self.posterior_log_variance_clipped}[self.model_var_type]
            log_var = extract(tab, t, x.shape)
        mt = self.model_mean_type
        if mt == ModelMeanType.PREVIOUS_X:
            return model_output, log_var
        if mt == ModelMeanType.START_X:
            pred = model_output
        elif mt == ModelMeanType.EPSILON:
            pred = (extract(self.sqrt_recip_alphas_cumprod, t, x.shape) * x
                    - extract(self.sqrt_recipm1_alphas_cumprod, t, x.shape) * model_output)
        elif mt == ModelMeanType.VELOCITY:
            # reference :394-399 gathers with t.shape, which cannot broadcast against an image batch
            raise RuntimeError("VELOCITY with a variational-bound term: the reference's _predict_xstart_from_v fails to broadcast")
        else:
            raise NotImplementedError(mt)
        mean, _, _ = self.q_posterior_mean_variance(pred, x, t)
        return mean, log_var

    # ---- sampling side (reference :278-384, :411-416, :461-560, :603-790) --------------------------------------
    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """{mean, variance, log_variance, pred_xstart} of p(x_{t-1} | x_t)."""
        model_kwargs = model_kwargs or {}
        B, C = x.shape[:2]
        assert t.shape == (B,)
        out = model(x, self._scale_timesteps(t), **model_kwargs)
        out = out[0] if isinstance(out, tuple) else out
        mean_out = out
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            assert out.shape == (B, C * 2, *x.shape[2:])
            mean_out = torch.split(out, C, dim=1)[0]
        _, log_var = self._model_log_variance(out, x, t)

        def process(v):
            if denoised_fn is not None:
                v = denoised_fn(v)
            return v.clamp(-1, 1) if clip_denoised else v

        mt = self.model_mean_type
        if mt == ModelMeanType.PREVIOUS_X:
            pred = process(extract(1.0 / self.posterior_mean_coef1, t, x.shape) * mean_out
                           - extract(self.posterior_mean_coef2 / self.posterior_mean_coef1, t, x.shape) * x)
            mean = mean_out
        elif mt in (ModelMeanType.START_X, ModelMeanType.EPSILON):
            if mt == ModelMeanType.START_X:
                pred = process(mean_out)
            else:
                pred = process(extract(self.sqrt_recip_alphas_cumprod, t, x.shape) * x
                               - extract(self.sqrt_recipm1_alphas_cumprod, t, x.shape) * mean_out)
            mean, _, _ = self.q_posterior_mean_variance(pred, x, t)
        elif mt == ModelMeanType.VELOCITY:
            raise RuntimeError("VELOCITY: the reference's _predict_xstart_from_v fails to broadcast (:394-399)")
        else:
            raise NotImplementedError(mt)
        assert mean.shape == log_var.shape == pred.shape == x.shape
        return {"mean": mean, "variance": torch.exp(log_var), "log_variance": log_var, "pred_xstart": pred}

    def _model_log_variance(self, model_output, x, t):
        """(mean half of the output, log variance) -- the variance branch of p_mean_variance (:304-330)."""
        B, C = x.shape[:2]
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            model_output, var_values = torch.split(model_output, C, dim=1)
            if self.model_var_type == ModelVarType.LEARNED:
                return model_output, var_values
            min_log = extract(self.posterior_log_variance_clipped, t, x.shape)
            max_log = extract(np.log(self.betas), t, x.shape)
            frac = (var_values + 1) / 2
            return model_output, frac * max_log + (1 - frac) * min_log
        tab = {ModelVarType.FIXED_LARGE: np.log(np.append(self.posterior_variance[1], self.betas[1:])),
               ModelVarType.FIXED_SMALL: self.posterior_log_variance_clipped}[self.model_var_type]
        return model_output, extract(tab, t, x.shape)

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return ((extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart)
                / extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        noise = torch.randn_like(x)
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        sample = out["mean"] + nonzero_mask * torch.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None, eta=0.0):
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        eps = self._predict_eps_from_xstart(x, t, out["pred_xstart"])
        alpha_bar = extract(self.alphas_cumprod, t, x.shape)
        alpha_bar_prev = extract(self.alphas_cumprod_prev, t, x.shape)
        sigma = eta * torch.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar)) * torch.sqrt(1 - alpha_bar / alpha_bar_prev)
        noise = torch.randn_like(x)
        mean_pred = out["pred_xstart"] * torch.sqrt(alpha_bar_prev) + torch.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        return {"sample": mean_pred + nonzero_mask * sigma * noise, "pred_xstart": out["pred_xstart"]}

    def _loop(self, step, model, shape, noise, device, **kw):
        img = noise if noise is not None else torch.randn(*shape, device=device)
        for i in list(range(self.num_timesteps))[::-1]:
            t = torch.tensor([i] * shape[0], device=device)
            with torch.no_grad():
                out = step(model, img, t, **kw)
                yield out
                img = out["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                                  device="cpu"):
        return self._loop(self.p_sample, model, shape, noise, device, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                          model_kwargs=model_kwargs)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, model_kwargs=None,
                                     device="cpu", eta=0.0):
        return self._loop(self.ddim_sample, model, shape, noise, device, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                          model_kwargs=model_kwargs, eta=eta)

    def p_sample_loop(self, model, shape, **kw):
        final = None
        for final in self.p_sample_loop_progressive(model, shape, **kw):
            pass
        return final["sample"]

    def ddim_sample_loop(self, model, shape, **kw):
        final = None
        for final in self.ddim_sample_loop_progressive(model, shape, **kw):
            pass
        return final["sample"]

    def _vb_terms_bpd(self, model_output, x_start, x_t, t):
        """[N] bits per dim: KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) for t > 0, the decoder NLL at t = 0."""
        true_mean, _, true_log_var = self.q_posterior_mean_variance(x_start, x_t, t)
        mean, log_var = self.model_mean_log_variance(model_output, x_t, t)
        kl = mean_flat(normal_kl(true_mean, true_log_var, mean, log_var)) / np.log(2.0)
        nll = -discretized_gaussian_log_likelihood(x_start, means=mean, log_scales=0.5 * log_var)
        nll = mean_flat(nll) / np.log(2.0)
        return torch.where(t == 0, nll, kl)


class FlowMatching:
    """reference :1151-1340, training side only."""

    def __init__(self, *, args, model_mean_type, device="cpu"):
        self.args = args
        self.model_mean_type = model_mean_type
        self.mse_loss_weight_type = args.weight_type
        self.path_type = args.path_type
        self.p2_gamma = args.p2_gamma
        self.p2_k = args.p2_k
        self.gamma = args.gamma
        self.learn_sigma = args.learn_sigma

    @staticmethod
    def expand_t_like_x(t, x):
        if t.dim() == 0:
            t = t.expand(x.shape[0])
        return t.view(t.size(0), *([1] * (x.dim() - 1))).to(x)

    def interpolant(self, t):
        p = self.path_type
        if p == "linear":
            return 1 - t, t, torch.full_like(t, -1.0), torch.full_like(t, 1.0)
        if p == "cosine":
            h = t * np.pi / 2
            return torch.cos(h), torch.sin(h), -np.pi / 2 * torch.sin(h), np.pi / 2 * torch.cos(h)
        if p == "linear_logsnr":
            lam = 10 + t * (-10.0 - 10)
            a, s = torch.sigmoid(0.5 * lam), torch.sigmoid(-0.5 * lam)
            da = -10.0 * a * s
            return a, s, da, -da
        raise NotImplementedError()

    def sample_t(self, x_start):
        td = self.args.time_dist
        n = x_start.shape[0]
        if td[0] == "uniform":
            return torch.rand(n, device=x_start.device)
        if td[0] == "lognorm":
            mu, sd = float(td[-2]), float(td[-1])
            return torch.sigmoid(torch.randn(n, device=x_start.device) * sd + mu)
        raise NotImplementedError(f"Unknown time_dist: {td}")

    def q_sample(self, x_start, noise, t):
        a, s, _, _ = self.interpolant(self.expand_t_like_x(t, x_start))
        return a * x_start + s * noise

    def compute_target(self, x_start, noise, t, alpha_t=None, sigma_t=None, d_alpha_t=None, d_sigma_t=None):
        if alpha_t is None or sigma_t is None or d_alpha_t is None or d_sigma_t is None:
            alpha_t, sigma_t, d_alpha_t, d_sigma_t = self.interpolant(t)
        e = lambda v: self.expand_t_like_x(v, x_start)
        mt = self.model_mean_type
        if mt == ModelMeanType.START_X:
            return x_start
        if mt == ModelMeanType.EPSILON:
            return noise
        if mt == ModelMeanType.VELOCITY:
            return e(alpha_t) * noise - e(sigma_t) * x_start
        if mt == ModelMeanType.VECTOR:
            return e(d_alpha_t) * x_start + e(d_sigma_t) * noise
        if mt == ModelMeanType.SCORE:
            return -noise / e(sigma_t)
        raise KeyError(mt)

    def training_losses(self, model, x_start, features=None, t=None, model_kwargs=None, noise=None):
        model_kwargs = model_kwargs or {}
        if noise is None:
            noise = torch.randn_like(x_start)
        if t is None:
            t = self.sample_t(x_start)
        a, s, da, ds = self.interpolant(t)
        x_t = self.q_sample(x_start, noise, t)
        w = compute_mse_loss_weight(self.model_mean_type, self.mse_loss_weight_type, t, a, s, self.p2_k, self.p2_gamma)
        target = self.compute_target(x_start, noise, t, a, s, da, ds)
        raw = model(x_t, t, **model_kwargs)
        out = raw[0] if isinstance(raw, tuple) else raw
        assert out.shape == target.shape == x_start.shape
        terms = {"mse": w * mean_flat((target - out) ** 2)}
        terms["loss"] = terms["mse"]
        return terms
