from .models import (ARCHS, AutoencoderKLEncoder, ControlNetModel, DDIMScheduler, SDArch, UNet2DConditionModel,
                     arch_for)
from .layers import PaddedContext
from .loading import load_component
