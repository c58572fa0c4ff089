from .net_factory import net_factory  # noqa: F401
from .unet import UNet, UNet_CCT  # noqa: F401
