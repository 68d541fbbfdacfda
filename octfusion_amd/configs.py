"""U-Net hyper-parameters of the reference's three diffusion configs, restated as data
(values from reference configs/octfusion_snet_uncond.yaml:8-27,
configs/octfusion_snet_cond.yaml:8-28, configs/octfusion_obja_uncond.yaml:8-27)."""
import copy

_COMMON = dict(full_depth=4, num_heads=4, dims=3)

SNET_UNCOND = dict(_COMMON, image_size=[16, 64], input_depth=[4, 6], unet_type=['lr', 'hr'],
                   df_type=['x0', 'eps'], input_channels=[8, 3], out_channels=[8, 3],
                   model_channels=[64, 128], num_res_blocks=[[1, 1, 1], [1, 1, 0]],
                   attention_resolutions=[2, 4], channel_mult=[[1, 2, 4], [1, 2, 4]],
                   use_checkpoint=False)

SNET_COND = dict(_COMMON, image_size=[16, 64], input_depth=[4, 6], unet_type=['lr', 'hr'],
                 df_type=['x0', 'eps'], input_channels=[8, 3], out_channels=[8, 3],
                 model_channels=[64, 128], num_res_blocks=[[1, 1, 1], [2, 2, 0]],
                 attention_resolutions=[2, 4, 8], channel_mult=[[1, 2, 4, 8], [1, 2, 4]],
                 use_checkpoint=False, num_classes=5)

OBJA_UNCOND = dict(_COMMON, image_size=[16, 64, 256], input_depth=[4, 6, 8],
                   unet_type=['lr', 'hr', 'feature'], df_type=['x0', 'x0', 'x0'],
                   input_channels=[8, 8, 3], out_channels=[8, 8, 3], model_channels=[64, 128, 64],
                   num_res_blocks=[[1, 1, 1], [2, 2, 0], [1, 1, 1]], attention_resolutions=[2, 4],
                   channel_mult=[[1, 2, 4], [1, 2, 4], [1, 2, 4]], use_checkpoint=True)

CONFIGS = {'snet_uncond': SNET_UNCOND, 'snet_cond': SNET_COND, 'obja_uncond': OBJA_UNCOND}


def unet_params(name, stage_flag):
    """kwargs for graph_unet_union.UNet3DModel."""
    p = copy.deepcopy(CONFIGS[name])
    p.pop('df_type')
    p['stage_flag'] = stage_flag
    return p


def stage_cfgs(name):
    """Per-stage dicts in the shape the oracle's functional U-Nets take."""
    p = CONFIGS[name]
    out = {}
    for i, kind in enumerate(p['unet_type']):
        if kind == 'lr':
            out[kind] = dict(kind='lr', full_depth=p['full_depth'], model_channels=p['model_channels'][i],
                             channel_mult=p['channel_mult'][i], attention_resolutions=p['attention_resolutions'],
                             num_heads=p['num_heads'], num_classes=p.get('num_classes'))
        else:
            out[kind] = dict(kind='hr', input_depth=p['input_depth'][i], full_depth=p['full_depth'],
                             model_channels=p['model_channels'][i], channel_mult=p['channel_mult'][i],
                             num_res_blocks=p['num_res_blocks'][i], num_classes=p.get('num_classes'))
    return out


# GraphVAE of each diffusion config (reference configs/vae_snet_eval.yaml:8-20 for ShapeNet;
# configs/vae_obja_eval_depth864.yaml:8-20 is the depth-8 / stop-6 Objaverse VAE the depth-8/6/4 cascade decodes with)
_VAE = dict(depth=8, channel_in=4, nout=4, full_depth=4, depth_stop=6, depth_out=8, resblk_type='basic', bottleneck=4,
            resblk_num=2, code_channel=16, embed_dim=3)
VAES = {'snet_uncond': _VAE, 'snet_cond': _VAE, 'obja_uncond': _VAE}


def vae_params(name):
    """kwargs for graph_vae.GraphVAE."""
    return dict(VAES[name])
