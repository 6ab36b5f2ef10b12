"""medsam/cfg.py drop-in: the same flags (reference: medsam/cfg.py:4-58). Only -mod, -mid_dim,
-image_size and -multimask_output influence the MFNet hot path; the rest are accepted for
command-line compatibility. Unknown flags of the HOST process are ignored instead of aborting
(the reference parses sys.argv inside the model constructor, unetformer_mmsam.py:487)."""
import argparse


def build_parser():
    p = argparse.ArgumentParser()
    a = p.add_argument
    a('-net', type=str, default='sam'); a('-baseline', type=str, default='unet'); a('-encoder', type=str, default='default')
    a('-seg_net', type=str, default='transunet')
    a('-mod', type=str, default='sam_lora', help='sam_lora (MMLoRA) | sam_adpt (MMAdapter)')
    a('-exp_name', default='msa_test_isic', type=str); a('-type', type=str, default='map'); a('-vis', type=int, default=None)
    a('-reverse', type=bool, default=False); a('-pretrain', type=bool, default=False); a('-val_freq', type=int, default=5)
    a('-gpu', type=bool, default=True); a('-gpu_device', type=int, default=0); a('-sim_gpu', type=int, default=0)
    a('-epoch_ini', type=int, default=1); a('-image_size', type=int, default=1024); a('-out_size', type=int, default=256)
    a('-patch_size', type=int, default=2); a('-dim', type=int, default=512); a('-depth', type=int, default=1)
    a('-heads', type=int, default=16); a('-mlp_dim', type=int, default=1024); a('-w', type=int, default=4)
    a('-b', type=int, default=2); a('-s', type=bool, default=True); a('-warm', type=int, default=1)
    a('-lr', type=float, default=1e-4); a('-uinch', type=int, default=1); a('-imp_lr', type=float, default=3e-4)
    a('-weights', type=str, default=0); a('-base_weights', type=str, default=0); a('-sim_weights', type=str, default=0)
    a('-distributed', default='none', type=str); a('-dataset', default='isic', type=str); a('-sam_ckpt', default=None)
    a('-thd', type=bool, default=False); a('-chunk', type=int, default=None); a('-num_sample', type=int, default=4)
    a('-roi_size', type=int, default=96); a('-evl_chunk', type=int, default=None)
    a('-mid_dim', type=int, default=None, help='middle dim of adapter or the rank of lora matrix')
    a('-multimask_output', type=int, default=1); a('-data_path', type=str, default='../data')
    return p


def parse_args(argv=None):
    opt, _unknown = build_parser().parse_known_args(argv)
    return opt
