"""Variance-aware-weighted diffusion objective on HIP kernels; same call surface as the reference's
`tools/gaussian_diffusion.py` (enums :21-56, schedules :59-123, GaussianDiffusion :126-930,
compute_mse_loss_weight :1092-1148, FlowMatching :1151-1340 training side).

What runs where
  host, once      float64 schedule tables (numpy, as the reference), and from them f32 device tables of
                  sqrt(abar_t), sqrt(1-abar_t), the loss weight w_t and the target coefficients -- the
                  reference re-uploads a float64 table on each of its 4-6 `_extract_into_tensor` calls per step
  vaw_qsample_fwd x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps, table gather fused        (12 B/element)
  vaw_wmse_fwd    target + (target-out)^2 + mean over CHW + per-sample weight, one pass  (12 B/element)
  vaw_wmse_bwd    d(out) in one pass                                                       (16 B/element)
Only the selected target is computed (the reference evaluates all four, :823-830).
"""
import enum
import math

import numpy as np
import torch

from . import ops


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


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)], dtype=np.float64)


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, lambda_max=10.0, lambda_min=-10.0):
    n = num_diffusion_timesteps
    if schedule_name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(n, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    if schedule_name == "linear_logsnr":
        return betas_for_alpha_bar(n, lambda t: 1.0 / (1.0 + math.exp(-(lambda_max + t * (lambda_min - lambda_max)))))
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    res = torch.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)


def compute_mse_loss_weight(model_mean_type, mse_loss_weight_type, t, alpha, sigma, p2_k=1.0, p2_gamma=1.0):
    """Per-sample loss weight (the paper's variance-aware weight is 'lambda').  [B]-sized torch arithmetic,
    kept op-for-op like the reference (:1092-1148) -- including that the result may alias `sigma` and that
    the snr==0 patch writes in place -- because the device weight TABLE is built by running exactly this on
    t = arange(T)."""
    snr = (alpha / sigma) ** 2
    wt = mse_loss_weight_type
    if wt == "constant":
        return torch.ones_like(t)
    kind = model_mean_type.name
    w = None

    def clip_snr(prefix, fn):
        k = float(wt.split(prefix)[-1])
        return fn(torch.stack([snr, k * torch.ones_like(t)], dim=1), dim=1)[0]

    if kind == "EPSILON":
        if wt.startswith("min_snr_"):
            w = clip_snr("min_snr_", torch.min) / snr
        elif wt.startswith("max_snr_"):
            w = clip_snr("max_snr_", torch.max) / snr
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
    elif kind == "START_X":
        if wt == "trunc_snr":
            w = torch.stack([snr, torch.ones_like(t)], dim=1).max(dim=1)[0]
        elif wt == "snr":
            w = snr
        elif wt == "inv_snr":
            w = 1.0 / snr
        elif wt.startswith("min_snr_"):
            w = clip_snr("min_snr_", torch.min)
        elif wt.startswith("max_snr_"):
            w = clip_snr("max_snr_", torch.max)
        elif wt == "lambda":
            w = alpha
    elif kind == "VECTOR":
        if wt == "lambda":
            w = torch.ones_like(t)
    elif kind == "VELOCITY":
        if wt.startswith("min_snr_"):
            w = clip_snr("min_snr_", torch.min) / (snr + 1)
        elif wt == "lambda":
            w = alpha * sigma
    if w is None:
        raise ValueError(f"Invalid mse_loss_weight_type: {wt}")
    w[snr == 0] = 1.0
    return w


class GaussianDiffusion:
    def __init__(self, *, args, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False,
                 device="cuda"):
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

        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas >= 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        self.alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(self.alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(self.alphas) / (1.0 - self.alphas_cumprod)
        self._dev = {}

    # ---- device tables ---------------------------------------------------------------------
    def _tables(self, device):
        key = str(device)
        tb = self._dev.get(key)
        if tb is None:
            a = torch.from_numpy(self.sqrt_alphas_cumprod).float()       # == from_numpy(f64)[t].float()
            s = torch.from_numpy(self.sqrt_one_minus_alphas_cumprod).float()
            t_all = torch.arange(self.num_timesteps)
            w = compute_mse_loss_weight(self.model_mean_type, self.mse_loss_weight_type, t_all, a.clone(), s.clone(),
                                        self.p2_k, self.p2_gamma).float()
            mt = self.model_mean_type
            one, zero = torch.ones_like(a), torch.zeros_like(a)
            if mt == ModelMeanType.EPSILON:
                ca, cb = zero, one
            elif mt == ModelMeanType.START_X:
                ca, cb = one, zero
            elif mt == ModelMeanType.VELOCITY:
                # the reference hands compute_target the sigma that the weight function may have patched in place
                # (alias, :1107/:1147); only EPSILON+'lambda' aliases and that never reaches this branch
                ca, cb = -s, a
            elif mt == ModelMeanType.PREVIOUS_X:
                # posterior mean of q(x_{t-1}|x_t,x_0) with x_t = a x0 + s eps  (:263-266)
                c1 = torch.from_numpy(self.posterior_mean_coef1).float()
                c2 = torch.from_numpy(self.posterior_mean_coef2).float()
                ca, cb = c1 + c2 * a, c2 * s
            else:
                raise KeyError(mt)
            tb = {k: v.contiguous().to(device) for k, v in dict(a=a, s=s, w=w, ca=ca, cb=cb).items()}
            tb["vb"] = self._vb_table().to(device)
            self._dev[key] = tb
        return tb

    def _vb_table(self):
        """[T, 8] f32 rows for vaw_vb_fwd: the float64 tables of q_posterior_mean_variance / p_mean_variance
        (:254-276, :304-330, :386-392) cast to f32 exactly as _extract_into_tensor does."""
        f = lambda arr: torch.from_numpy(np.asarray(arr, dtype=np.float64)).float()
        T = self.num_timesteps
        vt, mt = self.model_var_type, self.model_mean_type
        if vt == ModelVarType.FIXED_LARGE:
            lv_aux = f(np.log(np.append(self.posterior_variance[1], self.betas[1:])))
        elif vt == ModelVarType.FIXED_SMALL:
            lv_aux = f(self.posterior_log_variance_clipped)
        else:
            lv_aux = f(np.log(self.betas))
        if mt == ModelMeanType.EPSILON:
            pa, pb = f(self.sqrt_recip_alphas_cumprod), -f(self.sqrt_recipm1_alphas_cumprod)
        else:                                   # START_X: pred = model output; PREVIOUS_X ignores pa/pb
            pa, pb = torch.zeros(T), torch.ones(T)
        t0 = torch.zeros(T)
        t0[0] = 1.0
        return torch.stack([f(self.posterior_mean_coef1), f(self.posterior_mean_coef2), f(self.posterior_log_variance_clipped),
                            lv_aux, pa, pb, t0, torch.zeros(T)], dim=1).contiguous()

    # ---- sampling side (reference :278-384, :461-601, :603-790; no denoised_fn / cond_fn) ----------------------
    def _sample_table(self):
        """[T, 16] f32 rows for vaw_sample_step.  Every entry is the reference's float64 table cast to f32 as
        _extract_into_tensor does; products of tables (ddim sigma) are formed in f32 in the kernel, as the reference's
        tensor ops do."""
        f = lambda arr: torch.from_numpy(np.asarray(arr, dtype=np.float64)).float()
        T = self.num_timesteps
        vt, mt = self.model_var_type, self.model_mean_type
        if vt == ModelVarType.FIXED_LARGE:
            lv_aux = f(np.log(np.append(self.posterior_variance[1], self.betas[1:])))
        elif vt == ModelVarType.FIXED_SMALL:
            lv_aux = f(self.posterior_log_variance_clipped)
        else:
            lv_aux = f(np.log(self.betas))
        ra, rm1 = f(self.sqrt_recip_alphas_cumprod), f(self.sqrt_recipm1_alphas_cumprod)
        if mt == ModelMeanType.EPSILON:
            pa, pb = ra, -rm1
        elif mt == ModelMeanType.PREVIOUS_X:                      # (xprev - coef2*x_t) / coef1  (:401-409)
            pa, pb = -f(self.posterior_mean_coef2 / self.posterior_mean_coef1), f(1.0 / self.posterior_mean_coef1)
        else:
            pa, pb = torch.zeros(T), torch.ones(T)
        ab, abp = f(self.alphas_cumprod), f(self.alphas_cumprod_prev)
        s1, s2 = torch.sqrt((1 - abp) / (1 - ab)), torch.sqrt(1 - ab / abp)      # ddim sigma = eta * s1 * s2 (:636-640)
        t0 = torch.zeros(T)
        t0[0] = 1.0
        z = torch.zeros(T)
        return torch.stack([pa, pb, f(self.posterior_mean_coef1), f(self.posterior_mean_coef2),
                            f(self.posterior_log_variance_clipped), lv_aux, ra, rm1, torch.sqrt(abp), s1, abp, t0, s2, z, z, z],
                           dim=1).contiguous()

    def _reverse_step(self, kind, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=0.0, want_all=False):
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn (classifier guidance) are out of scope (SURVEY.md §2)")
        mt, vt = self.model_mean_type, self.model_var_type
        if mt == ModelMeanType.VELOCITY:
            raise RuntimeError("VELOCITY: the reference's _predict_xstart_from_v fails to broadcast (:394-399)")
        if mt not in (ModelMeanType.EPSILON, ModelMeanType.START_X, ModelMeanType.PREVIOUS_X):
            raise NotImplementedError(mt)
        B, C = x.shape[:2]
        assert t.shape == (B,)
        out = model(x, self._scale_timesteps(t), **(model_kwargs or {}))
        out = out[0] if isinstance(out, tuple) else out
        var_out = None
        if vt in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            assert out.shape == (B, C * 2, *x.shape[2:])
            out, var_out = torch.split(out, C, dim=1)
        assert out.shape == x.shape
        key = "ss"
        tb = self._tables(x.device)
        if key not in tb:
            tb[key] = self._sample_table().to(x.device)
        noise = None
        if kind:
            if getattr(self.args, "cpu_rng", False):
                noise = torch.randn(x.shape, dtype=torch.float32).to(x.device)     # the reference's CPU stream (parity runs)
            else:
                noise = torch.randn_like(x)
        var_mode = {ModelVarType.LEARNED: 1, ModelVarType.LEARNED_RANGE: 2}.get(vt, 0)
        return ops.sample_step(kind, out, var_out, x, noise, tb[key][t], 1 if mt == ModelMeanType.PREVIOUS_X else 0, var_mode,
                               clip_denoised, eta, want_all)

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        r = self._reverse_step(0, model, x, t, clip_denoised, denoised_fn, None, model_kwargs, want_all=True)
        return {"mean": r["mean"], "variance": torch.exp(r["log_variance"]), "log_variance": r["log_variance"],
                "pred_xstart": r["pred_xstart"]}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        return self._reverse_step(1, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        return self._reverse_step(2, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=eta)

    def _loop(self, step, model, shape, noise, device, progress, **kw):
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        if noise is not None:
            img = noise
        elif getattr(self.args, "cpu_rng", False):
            img = torch.randn(*shape).to(device)
        else:
            img = torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        for i in indices:
            t = torch.full((shape[0],), i, device=device, dtype=torch.long)
            with torch.no_grad():
                out = step(model, img, t, **kw)
                yield out
                img = out["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        return self._loop(self.p_sample, model, shape, noise, device, progress, clip_denoised=clip_denoised,
                          denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs)

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False):
        final = None
        for final in self.p_sample_loop_progressive(model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                                                    device, progress):
            pass
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0):
        return self._loop(self.ddim_sample, model, shape, noise, device, progress, clip_denoised=clip_denoised,
                          denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs, eta=eta)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                         device=None, progress=False, eta=0.0):
        final = None
        for final in self.ddim_sample_loop_progressive(model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                                                       device, progress, eta):
            pass
        return final["sample"]

    def _vb_terms_bpd(self, mean_out, var_out, x_start, x_t, t, scale=1.0):
        """reference :775-808 on the fused kernel.  mean_out / var_out: the two halves of the model output."""
        mt, vt = self.model_mean_type, self.model_var_type
        if mt == ModelMeanType.VELOCITY:
            # the reference gathers with t.shape in _predict_xstart_from_v (:394-399) and cannot broadcast
            raise RuntimeError("VELOCITY with a variational-bound term: the reference's _predict_xstart_from_v fails to broadcast")
        if mt not in (ModelMeanType.EPSILON, ModelMeanType.START_X, ModelMeanType.PREVIOUS_X):
            raise NotImplementedError(mt)
        var_mode = {ModelVarType.LEARNED: 1, ModelVarType.LEARNED_RANGE: 2}.get(vt, 0)
        coef = self._tables(x_start.device)["vb"][t]
        return ops.vb_terms(mean_out, var_out, x_start, x_t, coef, 1 if mt == ModelMeanType.PREVIOUS_X else 0, var_mode, scale)

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        tb = self._tables(x_start.device)
        return ops.qsample(x_start.contiguous(), noise.contiguous(), t, tb["a"], tb["s"])

    def sample_t(self, x_start):
        if self.args.time_dist[0] == "uniform":
            return torch.randint(0, self.num_timesteps, (x_start.shape[0],), device=x_start.device)
        raise NotImplementedError(f"Unknown time_dist: {self.args.time_dist}")

    def compute_target(self, x_start, noise, t, alpha=None, sigma=None):
        tb = self._tables(x_start.device)
        if self.model_mean_type == ModelMeanType.START_X:
            return x_start
        if self.model_mean_type == ModelMeanType.EPSILON:
            return noise
        return ops.mix_rows(x_start.contiguous(), noise.contiguous(), tb["ca"][t], tb["cb"][t])

    def training_losses(self, model, x_start, features=None, t=None, model_kwargs=None, noise=None):
        if model_kwargs is None:
            model_kwargs = {}
        if noise is None:
            noise = torch.randn_like(x_start)       # drawn BEFORE t, as the reference (:849-852)
        if t is None:
            t = self.sample_t(x_start)
        kl_loss = self.loss_type in (LossType.KL, LossType.RESCALED_KL)
        if not kl_loss and self.loss_type not in (LossType.MSE, LossType.RESCALED_MSE):
            raise NotImplementedError(self.loss_type)
        if getattr(self.args, "learn_align", False):
            raise NotImplementedError("learn_align: feature-alignment teachers are out of scope (SURVEY.md §2.1 row 12)")
        x_start = x_start.contiguous()
        noise = noise.contiguous()
        tb = self._tables(x_start.device)
        x_t = ops.qsample(x_start, noise, t, tb["a"], tb["s"])
        raw_output = model(x_t, self._scale_timesteps(t), **model_kwargs)
        model_output = raw_output[0] if isinstance(raw_output, tuple) else raw_output
        learned = self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE)
        var_values = None
        if learned:
            B, C = x_t.shape[:2]
            assert model_output.shape == (B, C * 2, *x_t.shape[2:])
            model_output, var_values = torch.split(model_output, C, dim=1)
        if kl_loss:                                                           # reference :865-876
            scale = float(self.num_timesteps) if self.loss_type == LossType.RESCALED_KL else 1.0
            return {"loss": self._vb_terms_bpd(model_output, var_values, x_start, x_t, t, scale)}
        terms = {}
        if learned:                                                           # reference :887-906
            # the bound trains the variance only: the mean prediction enters it detached
            scale = self.num_timesteps / 1000.0 if self.loss_type == LossType.RESCALED_MSE else 1.0
            terms["vb"] = self._vb_terms_bpd(model_output.detach(), var_values, x_start, x_t, t, scale)
        assert model_output.shape == x_start.shape
        terms["mse"] = ops.weighted_mse(model_output, x_start, noise, tb["ca"][t], tb["cb"][t], tb["w"][t])
        terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
        return terms


class FlowMatching:
    """Continuous-time sibling (reference :1151-1340, training side): same kernels, coefficients from the
    interpolant instead of the tables."""

    def __init__(self, *, args, model_mean_type, device="cuda"):
        self.args = args
        self.model_mean_type = model_mean_type
        self.mse_loss_weight_type = args.weight_type
        self.path_type = args.path_type
        self.sampler_type = getattr(args, "sampler_type", "ode")
        self.p2_gamma = args.p2_gamma
        self.p2_k = args.p2_k
        self.gamma = args.gamma
        self.learn_sigma = args.learn_sigma

    def expand_t_like_x(self, t, x):
        if t.dim() == 0:
            t = t.expand(x.shape[0])
        return t.view(t.size(0), *([1] * (len(x.size()) - 1))).to(x)

    def interpolant(self, t):
        if self.path_type == "linear":
            return 1 - t, t, torch.full_like(t, -1.0), torch.full_like(t, 1.0)
        if self.path_type == "cosine":
            return (torch.cos(t * np.pi / 2), torch.sin(t * np.pi / 2), -np.pi / 2 * torch.sin(t * np.pi / 2),
                    np.pi / 2 * torch.cos(t * np.pi / 2))
        if self.path_type == "linear_logsnr":
            lam = 10 + t * (-10.0 - 10)
            alpha_t, sigma_t = torch.sigmoid(0.5 * lam), torch.sigmoid(-0.5 * lam)
            d_alpha_t = -10.0 * alpha_t * sigma_t
            return alpha_t, sigma_t, d_alpha_t, -d_alpha_t
        raise NotImplementedError()

    def sample_t(self, x_start):
        td = self.args.time_dist
        if td[0] == "uniform":
            return torch.rand(x_start.shape[0], device=x_start.device)
        if td[0] == "lognorm":
            mu, sigma = float(td[-2]), float(td[-1])
            return torch.sigmoid(torch.randn(x_start.shape[0], device=x_start.device) * sigma + mu)
        raise NotImplementedError(f"Unknown time_dist: {td}")

    def q_sample(self, x_start, noise, t):
        alpha_t, sigma_t, _, _ = self.interpolant(t.float())
        return ops.mix_rows(x_start.contiguous(), noise.contiguous(), alpha_t.contiguous(), sigma_t.contiguous())

    def _target_coefs(self, alpha_t, sigma_t, d_alpha_t, d_sigma_t):
        mt = self.model_mean_type
        one, zero = torch.ones_like(alpha_t), torch.zeros_like(alpha_t)
        if mt == ModelMeanType.START_X:
            return one, zero
        if mt == ModelMeanType.EPSILON:
            return zero, one
        if mt == ModelMeanType.VELOCITY:
            return -sigma_t, alpha_t
        if mt == ModelMeanType.VECTOR:
            return d_alpha_t, d_sigma_t
        if mt == ModelMeanType.SCORE:
            return zero, -1.0 / sigma_t
        raise KeyError(mt)

    def compute_target(self, x_start, noise, t, alpha_t=None, sigma_t=None, d_alpha_t=None, d_sigma_t=None):
        if alpha_t is None or sigma_t is None or d_alpha_t is None or d_sigma_t is None:
            alpha_t, sigma_t, d_alpha_t, d_sigma_t = self.interpolant(t)
        ca, cb = self._target_coefs(alpha_t, sigma_t, d_alpha_t, d_sigma_t)
        return ops.mix_rows(x_start.contiguous(), noise.contiguous(), ca.float().contiguous(), cb.float().contiguous())

    def training_losses(self, model, x_start, features=None, t=None, model_kwargs=None, noise=None):
        if model_kwargs is None:
            model_kwargs = {}
        if noise is None:
            noise = torch.randn_like(x_start)
        if t is None:
            t = self.sample_t(x_start)
        if getattr(self.args, "learn_align", False):
            raise NotImplementedError("learn_align: feature-alignment teachers are out of scope (SURVEY.md §2.1 row 12)")
        x_start, noise = x_start.contiguous(), noise.contiguous()
        alpha_t, sigma_t, d_alpha_t, d_sigma_t = self.interpolant(t)
        x_t = ops.mix_rows(x_start, noise, alpha_t.contiguous(), sigma_t.contiguous())
        # targets use the un-patched sigma for every type but the aliasing one; take them first
        ca, cb = self._target_coefs(alpha_t, sigma_t.clone(), d_alpha_t, d_sigma_t)
        w = compute_mse_loss_weight(self.model_mean_type, self.mse_loss_weight_type, t, alpha_t, sigma_t, self.p2_k,
                                    self.p2_gamma)
        raw_output = model(x_t, t, **model_kwargs)
        model_output = raw_output[0] if isinstance(raw_output, tuple) else raw_output
        assert model_output.shape == x_start.shape
        terms = {"mse": ops.weighted_mse(model_output, x_start, noise, ca.float().contiguous(), cb.float().contiguous(),
                                         w.float().contiguous())}
        terms["loss"] = terms["mse"]
        return terms
