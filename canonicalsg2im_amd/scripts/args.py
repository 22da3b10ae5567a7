"""Trainer flags of the hot path — same names and defaults as the reference parser
(`scripts/args.py:13-212`), restricted to the flags the path reads,
plus `init_args` (`scripts/args.py:215-238`) without its CUDA side effect."""
import argparse


def int_tuple(s):
    return tuple(int(i) for i in s.split(','))


def bool_flag(s):
    if s == '1':
        return True
    if s == '0':
        return False
    raise ValueError('Invalid value "%s" for bool flag (should be 0 or 1)' % s)


_OFF_PATH_FLAGS = {
    # data
    'img_deprocess': ('decode_img', 'str'), 'num_train_samples': (None, 'int'), 'num_val_samples': (1024, 'int'),
    'shuffle_val': (True, 'bool'), 'loader_num_workers': (1, 'int'), 'include_relationships': (True, 'bool'),
    'vg_image_dir': ('datasets/vg/images', 'str'), 'train_h5': ('datasets/vg/train.h5', 'str'),
    'val_h5': ('datasets/vg/val.h5', 'str'), 'vocab_json': ('datasets/vg/vocab.json', 'str'),
    'max_objects_per_image': (10, 'int'), 'vg_use_orphaned_objects': (True, 'bool'), 'dataroot': ('./datasets', 'str'),
    'preprocess_mode': ('scale_width_and_crop', 'str'), 'no_flip': (False, 'flag'), 'nThreads': (0, 'int'),
    'cache_filelist_write': (False, 'flag'), 'cache_filelist_read': (False, 'flag'), 'dense_scenes': (0, 'int'),
    'max_objects_val': (None, 'int'), 'min_object_size': (0.02, 'float'), 'use_attributes': (1, 'int'),
    'include_dummies': (0, 'int'), 'use_transitivity': (0, 'int'), 'all_transitive_baseline': (0, 'int'),
    'use_all_relations': (0, 'int'), 'learned_symmetry': (0, 'int'), 'use_converse': (0, 'int'),
    # run control, logging, checkpoints
    'timing': (False, 'bool'), 'checkpoint_every': (10000, 'int'), 'output_dir': (None, 'str'), 'run_name': ('debug', 'str'),
    'checkpoint_name': ('checkpoint', 'str'), 'checkpoint_gan_name': ('checkpoint', 'str'),
    'checkpoint_graph_name': ('checkpoint', 'str'), 'restore_checkpoint': (0, 'int'), 'checkpoint_start_from': (None, 'str'), 'name': ('label2coco', 'str'),
    'checkpoints_dir': ('./checkpoints', 'str'), 'phase': ('train', 'str'), 'load_from_opt_file': (False, 'flag'),
    'display_winsize': (400, 'int'), 'debug': (False, 'flag'), 'niter': (50, 'int'), 'niter_decay': (0, 'int'),
    'full_test': (1000000, 'int'), 'resolution': (256, 'int'),
    # model options the reference declares but its trainer never reads on this path
    'graph_model': ('jj', 'str'), 'heads': (1, 'int'), 'normalization': ('batch', 'str'), 'activation': ('leakyrelu-0.2', 'str'),
    'use_boxes_pred_after': (-1, 'int'), 'netD_subarch': ('n_layer', 'str'), 'model': ('pix2pix', 'str'),
    'norm_E': ('spectralinstance', 'str'), 'label_nc': (182, 'int'), 'contain_dontcare_label': (False, 'flag'),
    'output_nc': (131, 'int'), 'netG': ('spade', 'str'), 'init_type': ('xavier', 'str'), 'init_variance': (0.02, 'float'),
    'no_instance': (False, 'flag'), 'nef': (16, 'int'), 'optimizer': ('adam', 'str'), 'D_steps_per_G': (1, 'int'),
    'netD': ('multiscale', 'str'), 'no_TTUR': (False, 'flag'), 'lambda_kld': (0.05, 'float'), 'ndf_mask': (64, 'int'),
    'num_D_mask': (1, 'int'), 'norm_D_mask': ('instance', 'str'), 'n_layers_D_mask': (2, 'int'),
    'transformer_hidden_dim': (32, 'int'),
}


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', default='coco',
                   choices=['vg', 'clevr', 'coco', 'synthetic', 'packed_coco', 'packed_vg', 'packed_clevr'])
    # optimisation
    p.add_argument('--batch_size', default=4, type=int)
    p.add_argument('--num_iterations', default=1000000, type=int)
    p.add_argument('--learning_rate', default=1e-4, type=float)
    p.add_argument('--mask_learning_rate', default=1e-5, type=float)
    p.add_argument('--img_learning_rate', default=1e-4, type=float)
    p.add_argument('--beta1', default=0.5, type=float)
    p.add_argument('--image_size', default='256,256', type=int_tuple)
    # graph encoder
    p.add_argument('--mask_size', default=0, type=int)
    p.add_argument('--embedding_dim', default=32, type=int)
    p.add_argument('--gconv_dim', default=128, type=int)
    p.add_argument('--g_mask_dim', default=128 + 64, type=int)
    p.add_argument('--mask_noise_dim', default=64, type=int)
    p.add_argument('--gconv_hidden_dim', default=512, type=int)
    p.add_argument('--gconv_pooling', default='avg', type=str)
    p.add_argument('--gconv_num_layers', default=5, type=int)
    p.add_argument('--mlp_normalization', default='none', type=str)
    p.add_argument('--layout_noise_dim', default=32, type=int)
    p.add_argument('--learned_init', type=str, default='uniform', choices=['uniform', '0', '-4'])
    p.add_argument('--learned_transitivity', type=int, default=0)
    p.add_argument('--learned_converse', type=int, default=0)
    # generator / discriminator
    p.add_argument('--num_upsampling_layers', choices=('normal', 'more', 'most'), default='normal')
    p.add_argument('--ngf', type=int, default=64)
    p.add_argument('--num_D', type=int, default=2)
    p.add_argument('--n_layers_D', type=int, default=4)
    p.add_argument('--aspect_ratio', type=float, default=1.0)
    p.add_argument('--isTrain', default=1, type=int)
    p.add_argument('--use_vae', action='store_true')
    p.add_argument('--z_dim', type=int, default=256)
    p.add_argument('--norm_G', type=str, default='spectralspadesyncbatch3x3')
    p.add_argument('--norm_D', type=str, default='spectralinstance')
    p.add_argument('--ndf', type=int, default=64)
    p.add_argument('--rep_size', default=32, type=int)
    p.add_argument('--appearance_normalization', default='batch')
    p.add_argument('--a_activation', default='leakyrelu-0.2')
    p.add_argument('--pool_size', default=100, type=int)
    # losses
    p.add_argument('--no_ganFeat_loss', action='store_true')
    p.add_argument('--no_vgg_loss', action='store_true')
    p.add_argument('--gan_mode', type=str, default='hinge')
    p.add_argument('--gan_loss_type', default='gan')
    p.add_argument('--lambda_feat', type=float, default=10.0)
    p.add_argument('--lambda_vgg', type=float, default=10.0)
    p.add_argument('--lambda_obj', default=0.1, type=float)
    p.add_argument('--discriminator_img_loss_weight', default=1.0, type=float)
    p.add_argument('--discriminator_obj_loss_weight', default=0.1, type=float)
    p.add_argument('--discriminator_mask_loss_weight', default=1.0, type=float)
    p.add_argument('--bbox_pred_loss_weight', default=10, type=float)
    p.add_argument('--mask_pred_loss_weight', default=0, type=float)
    # object discriminator (next-row component; flags kept for the namespace)
    p.add_argument('--d_normalization', default='batch')
    p.add_argument('--d_padding', default='valid')
    p.add_argument('--d_activation', default='leakyrelu-0.2')
    p.add_argument('--d_obj_arch', default='C4-64-2,C4-128-2,C4-256-2')
    p.add_argument('--crop_size', default=32, type=int)
    p.add_argument('--ac_loss_weight', default=0.1, type=float)
    # switches
    p.add_argument('--skip_generation', type=int, default=0)
    p.add_argument('--skip_graph_model', type=int, default=0)
    p.add_argument('--use_img_disc', type=int, default=0)
    p.add_argument('--use_cuda', action='store_true')
    p.add_argument('--gpu_ids', type=str, default='0')
    p.add_argument('--freeze', default=0, type=int)
    p.add_argument('--freeze_options', default=None)
    p.add_argument('--print_every', default=10, type=int)
    p.add_argument('--min_objects', type=int)
    p.add_argument('--max_objects', type=int)
    # every other flag of the reference's command line (data loading, logging, checkpointing, evaluation, unused SPADE
    # options): accepted with the reference's defaults so that its recipes parse unchanged; none is read on the hot path
    for name, (default, kind) in _OFF_PATH_FLAGS.items():
        if kind == 'flag':
            p.add_argument('--' + name, action='store_true')
        else:
            p.add_argument('--' + name, default=default, type={'int': int, 'float': float, 'str': str, 'bool': bool_flag}[kind])
    return p


parser = build_parser()


def init_args(args):
    """`scripts/args.py:215-238`: parse gpu ids, check the batch divides, derive semantic_nc."""
    if isinstance(args.gpu_ids, str):
        args.gpu_ids = [int(s) for s in args.gpu_ids.split(',') if int(s) >= 0]
    assert len(args.gpu_ids) == 0 or args.batch_size % (len(args.gpu_ids)) == 0, \
        "Batch size %d is wrong. It must be a multiple of # GPUs %d." % (args.batch_size, len(args.gpu_ids))
    args.semantic_nc = len(args.vocab['attributes']) * args.embedding_dim
    return args


def make_opt(vocab, argv=(), **overrides):
    """Namespace exactly as the trainer builds it: parse, attach vocab, init_args."""
    args = build_parser().parse_args(list(argv))
    for k, v in overrides.items():
        setattr(args, k, v)
    args.vocab = vocab
    if isinstance(args.image_size, int):
        args.image_size = (args.image_size, args.image_size)
    return init_args(args)


def get_args():
    return parser.parse_args()
