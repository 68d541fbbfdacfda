"""Stage holder / dispatcher: builds unet_lr / unet_hr / unet_feature from the
list-valued YAML parameters and routes a call by ``unet_type``.

Mirror of reference models/networks/diffusion_networks/graph_unet_union.py
(:15-92).  The reference's ``unet_type == "lr"`` branch runs unet_lr a second
time with probability 0.5 and discards the result (it tests the key 'self_cond'
that no caller passes): outputs are unaffected, so the wasted
call is not reproduced.
"""
import torch.nn as nn

from . import graph_unet_hr, graph_unet_lr


class UNet3DModel(nn.Module):
    def __init__(self, stage_flag, image_size, input_depth, unet_type, full_depth, input_channels,
                 out_channels, model_channels, num_res_blocks, attention_resolutions, channel_mult,
                 num_heads, use_checkpoint, dims, num_classes=None, **kwargs):
        super().__init__()
        self.unet_lr = self.unet_hr = self.unet_feature = None
        for i, kind in enumerate(unet_type):
            if kind == 'lr':
                self.unet_lr = graph_unet_lr.UNet3DModel(
                    full_depth=full_depth, in_split_channels=input_channels[i],
                    model_channels=model_channels[i], out_split_channels=out_channels[i],
                    attention_resolutions=attention_resolutions, channel_mult=channel_mult[i],
                    use_checkpoint=use_checkpoint, num_heads=num_heads, dims=dims, num_classes=num_classes)
            elif kind in ('hr', 'feature'):
                net = graph_unet_hr.UNet3DModel(
                    image_size=image_size[i], input_depth=input_depth[i], full_depth=full_depth,
                    in_channels=input_channels[i], model_channels=model_channels[i],
                    lr_model_channels=model_channels[i - 1], out_channels=out_channels[i],
                    num_res_blocks=num_res_blocks[i], channel_mult=channel_mult[i], dims=dims,
                    use_checkpoint=use_checkpoint, num_heads=num_heads, num_classes=num_classes)
                if kind == 'hr':
                    self.unet_hr = net
                else:
                    self.unet_feature = net
            else:
                raise ValueError(kind)
            if kind == stage_flag:
                break

    wants_self_cond = False      # only the lr stage consumes x_self_cond, and it runs the x0 branch

    def forward(self, unet_type=None, **input_data):
        if unet_type == 'lr':
            return self.unet_lr(**input_data)
        if unet_type == 'hr':
            return self.unet_hr(**input_data)
        if unet_type == 'feature':
            return self.unet_feature(**input_data)
        raise ValueError(unet_type)
