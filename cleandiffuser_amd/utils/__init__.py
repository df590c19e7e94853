from .schedules import *          # noqa: F401,F403
from .schedules import (SUPPORTED_NOISE_SCHEDULES, SUPPORTED_DISCRETIZATIONS,
                        SUPPORTED_SAMPLING_STEP_SCHEDULE)
from .embeddings import (PositionalEmbedding, UntrainablePositionalEmbedding, SinusoidalEmbedding,
                         FourierEmbedding, UntrainableFourierEmbedding, SUPPORTED_TIMESTEP_EMBEDDING)
from .misc import (set_seed, at_least_ndim, to_tensor, count_parameters, ema_update, report_parameters, DD_RETURN_SCALE,
                   FreezeModules,
                   UnfreezeModules, EvalModules, TrainModules, dict_apply, loop_dataloader)
from .blocks import GroupNorm1d, Mlp
from .synth import synth_state_dict, load_synth, synth_array
from .critics import DQLCritic, TwinQ, V, IQL, IDQLQNet, IDQLVNet, SoftLowerBound, SoftUpperBound
from .normalizers import EmptyNormalizer, GaussianNormalizer, MinMaxNormalizer
from .misc import param_to_module, TensorDict, invalidate_weights
