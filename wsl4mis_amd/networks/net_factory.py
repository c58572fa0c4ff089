"""Drop-in for the reference's networks/net_factory.py:6-22 -- same name, arguments and return convention
(an nn.Module already on the GPU; unknown net types return None exactly like the reference)."""
from .unet import UNet, UNet_CCT

_NOT_BUILT = {"unet_cct_3h", "unet_ds", "efficient_unet", "pnet"}


def net_factory(net_type="unet", in_chns=1, class_num=3, conv_precision="f32"):
    """conv_precision (extension, keyword only in spirit): "f32" (default) or "split_f16x3" -- see networks/unet.py"""
    if net_type == "unet":
        return UNet(in_chns=in_chns, class_num=class_num, conv_precision=conv_precision)
    if net_type == "unet_cct":
        return UNet_CCT(in_chns=in_chns, class_num=class_num, conv_precision=conv_precision)
    if net_type in _NOT_BUILT:
        raise NotImplementedError(f"net_factory('{net_type}') exists in the reference but is outside the MI355X hot "
                                  "path built here (SURVEY.md section 2); 'unet' and 'unet_cct' are available")
    return None
