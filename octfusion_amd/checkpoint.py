"""On-disk formats of the reference, read and written without the reference.

* diffusion checkpoints `df_<label>.pth` (models/octfusion_model_union.py:501-545; 3-stage:
  models/octfusion_model_union_3t.py:219-229 save, :250-261 load):
  {'df_unet_lr', 'ema_df_unet_lr', ['df_unet_hr', 'ema_df_unet_hr'], ['df_unet_feature', 'ema_df_unet_feature'],
   'opt', 'global_step'};
* VAE checkpoints (models/model_utils.py:18-28): a state_dict, or {'autoencoder': sd}, or a
  '.solver.tar' file holding {'model_dict': sd};
* sample files written by tools/gen_split.py:46-54 and read by datasets/dualoctree_snet.py:137-151:
  `split_small.pth` = tensor [8, S, S, S] (batch dim squeezed), `split_large.pth` = tensor [nnum6, 8].
State-dict key names / shapes of octfusion_amd's modules equal the reference's (tests/test_gpu_parity.py), so
`load_state_dict(strict=True)` is the whole conversion.
"""
import os

import torch


def _read(ckpt, allow_pickle=False):
    """A checkpoint file -> its dict.  The formats above hold tensors, numbers and plain containers only, so the file is
    read with torch's restricted unpickler; allow_pickle=True (a file you trust that carries other Python objects, e.g.
    an option namespace) falls back to the full one."""
    if isinstance(ckpt, (str, os.PathLike)):
        return torch.load(ckpt, map_location='cpu', weights_only=not allow_pickle)
    return ckpt


STAGE_NETS = ('unet_lr', 'unet_hr', 'unet_feature')
_STAGES_UP_TO = {'lr': 1, 'hr': 2, 'feature': 3}


def load_ckpt(ckpt, df, ema_df=None, load_options=STAGE_NETS, opt=None, allow_pickle=False):
    """octfusion_model_union.py:525-545 / octfusion_model_union_3t.py:250-261.  df / ema_df: union UNet3DModel
    instances (ema_df may be None or df itself at inference); every stage net present in BOTH the file and `df` and
    named in load_options is loaded strictly.  opt: optional training.AdamW whose state is restored from 'opt' when
    the file holds one written by save_ckpt (a torch.optim state from the reference is keyed by parameter index and
    is not convertible without the reference's parameter order: ignored).  Returns the checkpoint's global_step."""
    sd = _read(ckpt, allow_pickle)
    for name in STAGE_NETS:
        if name in load_options and 'df_' + name in sd and getattr(df, name, None) is not None:
            getattr(df, name).load_state_dict(sd['df_' + name], strict=True)
            if ema_df is not None and getattr(ema_df, name, None) is not None:
                getattr(ema_df, name).load_state_dict(sd['ema_df_' + name], strict=True)
    if opt is not None and isinstance(sd.get('opt'), dict) and 'state' in sd['opt'] and 'param_groups' not in sd['opt']:
        opt.load_state_dict(sd['opt'])
    return sd.get('global_step')


def save_ckpt(path, df, ema_df, global_step, stage_flag='hr', opt_state=None):
    """octfusion_model_union.py:501-523 / octfusion_model_union_3t.py:219-229 (file layout only; rotation of old
    files is the trainer's business): the nets of every stage up to `stage_flag`."""
    sd = {'opt': opt_state if opt_state is not None else {}, 'global_step': global_step}
    for name in STAGE_NETS[:_STAGES_UP_TO[stage_flag]]:
        sd['df_' + name] = getattr(df, name).state_dict()
        sd['ema_df_' + name] = getattr(ema_df, name).state_dict()
    torch.save(sd, path)


def vae_state_dict(ckpt, allow_pickle=False):
    """models/model_utils.py:18-28: unwrap the three layouts a GraphVAE checkpoint comes in."""
    sd = _read(ckpt, allow_pickle)
    if isinstance(ckpt, (str, os.PathLike)) and str(ckpt).endswith('.solver.tar'):
        sd = sd['model_dict']
    if 'autoencoder' in sd:
        sd = sd['autoencoder']
    return sd


def load_vae(ckpt, vae, allow_pickle=False):
    vae.load_state_dict(vae_state_dict(ckpt, allow_pickle), strict=True)
    return vae.eval()


def read_splits(sample_dir, device):
    """A sample directory of the reference's dataset (dualoctree_snet.py:137-151) -> (split_small [1,8,S,S,S],
    split_large [nnum6, 8] or None) on `device`."""
    small = torch.load(os.path.join(sample_dir, 'split_small.pth'), map_location='cpu', weights_only=True)
    if small.dim() == 4:
        small = small.unsqueeze(0)                  # gen_split.py:52 squeezes the batch dim
    p = os.path.join(sample_dir, 'split_large.pth')
    large = torch.load(p, map_location='cpu', weights_only=True) if os.path.exists(p) else None
    return small.to(device), (large.to(device) if large is not None else None)


def write_splits(sample_dir, octree, full_depth=4, small_depth=6):
    """tools/gen_split.py:50-54 for one shape (octree.batch_size == 1)."""
    from .octree import octree2split_large, octree2split_small
    os.makedirs(sample_dir, exist_ok=True)
    torch.save(octree2split_small(octree, full_depth).squeeze(0).cpu(), os.path.join(sample_dir, 'split_small.pth'))
    if octree.depth > small_depth:
        torch.save(octree2split_large(octree, small_depth).cpu(), os.path.join(sample_dir, 'split_large.pth'))
