"""Public names of vaw_amd."""
from . import dist_util, gaussian_diffusion, ops, resample, respace, sampler, utils  # noqa: F401
from ._lib import LIB_PATH, VawError, exported_symbols, lib  # noqa: F401
from .dit import DiT, DiT_B, DiT_L, DiT_S, DiT_XL, DiT_models  # noqa: F401
from .flat import FlatModule  # noqa: F401
from .unet import (ADM_32, ADM_64, ADM_128, ADM_256, ADM_512, LDM, UNet_32, UNet_64, UNet_models, UNetModel,  # noqa: F401
                   create_unet_model)
from .gaussian_diffusion import (FlowMatching, GaussianDiffusion, LossType, ModelMeanType, ModelVarType,  # noqa: F401
                                 compute_mse_loss_weight, get_named_beta_schedule, mean_flat)
from .optim import FusedAdamW  # noqa: F401
from .parallel import DistributedDataParallel  # noqa: F401
from .data import DevicePrefetcher, LatentBatchLoader, LatentH5Dataset, ShardedSampler  # noqa: F401
from .samplers import EDMDenoiser, edm_sample, flow_ode_sample, flow_sde_sample  # noqa: F401
from .respace import SpacedDiffusion, space_timesteps  # noqa: F401
from .sampler import IntervalCFG  # noqa: F401
from .resample import (LossSecondMomentResampler, UniformSampler, create_named_schedule_sampler)  # noqa: F401
from .trainer import Trainer, ema, sample_from_latent  # noqa: F401
from .utils import get_lr_lambda, load_checkpoint, save_checkpoint, set_random_seed, warmup_cosine_lr  # noqa: F401
