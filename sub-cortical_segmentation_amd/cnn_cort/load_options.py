"""
Options loader with the reference's keys and string semantics
(mirrors /root/reference/cnn_cort/load_options.py:11-59; configuration.cfg:1-22).

Differences, all at the device line: the reference exports THEANO_FLAGS from `mode`
(load_options.py:53-57); here `mode` selects the HIP device instead:
    'cuda0', 'cuda1', ... / 'hip0', ...  -> device index N
    anything else (e.g. 'cpu')            -> device LOCAL_RANK (torchrun) or 0
There is no CPU execution path in this package.
"""
import configparser
import os
import re


def load_options(user_config):
    """Map a ConfigParser (sections [database] and [model]) to the flat options dict.
    Boolean-like options stay STRINGS ('True'/'False') exactly as in the reference, which compares them with
    == 'True' (nets.py:248, base.py:112,374,416,422,434,445,449)."""
    options = {}
    options['experiment'] = user_config.get('model', 'name')
    options['train_folder'] = user_config.get('database', 'train_folder')
    options['test_folder'] = user_config.get('database', 'inference_folder')
    options['output_folder'] = ''
    options['current_scan'] = ''
    options['t1_name'] = user_config.get('database', 't1_name')
    options['roi_name'] = user_config.get('database', 'roi_name')
    options['out_name'] = 'out_seg.nii.gz'
    options['save_tmp'] = user_config.get('database', 'save_tmp')

    options['mode'] = user_config.get('model', 'mode').strip()
    options['patch_size'] = [user_config.getint('model', 'patch_size'), user_config.getint('model', 'patch_size')]
    options['weight_paths'] = None
    options['train_split'] = user_config.getfloat('model', 'train_split')
    options['max_epochs'] = user_config.getint('model', 'max_epochs')
    options['patience'] = user_config.getint('model', 'patience')
    options['batch_size'] = user_config.getint('model', 'batch_size')
    options['test_batch_size'] = user_config.getint('model', 'test_batch_size')
    options['net_verbose'] = user_config.getint('model', 'net_verbose')
    options['load_weights'] = user_config.get('model', 'load_weights').strip()
    options['randomize_train'] = True
    options['debug'] = user_config.get('model', 'debug').strip()
    options['out_probabilities'] = user_config.get('model', 'out_probabilities').strip()
    options['post_process'] = user_config.get('model', 'post_process').strip()
    options['crop'] = user_config.get('model', 'speedup_segmentation').strip()
    options['device'] = device_from_mode(options['mode'])
    # optional keys the reference's configuration.cfg does not have
    if user_config.has_option('model', 'augment'):
        options['augment'] = user_config.get('model', 'augment').strip()
    if user_config.has_option('model', 'seed'):
        options['seed'] = user_config.getint('model', 'seed')
    if user_config.has_option('model', 'fwd_split'):
        options['fwd_split'] = user_config.get('model', 'fwd_split').strip()      # 'True': inference on the split kernels (nets.py)
    if user_config.has_option('model', 'fit_split'):
        options['fit_split'] = user_config.get('model', 'fit_split').strip()      # 'True': fit()'s forward conv1 + conv2 block on the split kernels
    return options


def device_from_mode(mode):
    m = re.match(r'^(?:cuda|hip|gpu)(\d+)$', str(mode).strip())
    if m:
        return int(m.group(1))
    return int(os.environ.get('LOCAL_RANK', '0'))


def read_config(path):
    """Convenience: parse an INI file the way train_model.py:23-26 does."""
    cfg = configparser.RawConfigParser()
    if not cfg.read(path):
        raise IOError('cannot read configuration file %s' % path)
    return load_options(cfg)


def print_options(options):
    """load_options.py:62-72."""
    print("--------------------------------------------------")
    print(" ")
    for k in options.keys():
        print(k, ':', options[k])
    print("--------------------------------------------------")


def is_true(v):
    """The reference's truthiness for string options: == 'True'; real booleans are accepted too."""
    return v is True or (isinstance(v, str) and v.strip() == 'True')
