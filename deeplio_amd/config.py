"""Default configuration dictionary (the keys train.py/test.py load from config.yaml in the
reference: config.yaml:1-132) and helpers to specialise it.  `load_config(path)` reads a
user YAML with the same keys."""
import copy


def default_config():
    return {
        'datasets': {
            'sequence-size': 5,
            'combinations': [[0, 1], [1, 2], [2, 3], [3, 4], [4, 5]],
            'kitti': {
                'root-path-sync': "datasets/KITTI/sync", 'root-path-unsync': "datasets/KITTI/extract",
                'image-width': 720, 'image-height': 57, 'crop-factors': [0, 0],
                'fov-up': 3., 'fov-down': -25., 'max-depth': 80., 'min-depth': 1.,
                'inverse-depth': True,
                'mean-image': [-0.0014, 0.0043, -0.011, 0.2258, -0.0024, 0.0037, 0.3793, 0.1115],
                'std-image': [0.1269, 0.0951, 0.0108, 0.1758, 0.3436, 0.4445, 0.5664, 0.0884],
                'mean-imu': [-0.0685, 0.1672, 9.7967, -0., 0.0006, 0.0059],
                'std-imu': [0.8766, 0.9528, 0.3471, 0.0204, 0.0227, 0.1412],
            },
        },
        'deeplio': {
            'dropout': 0.25, 'pretrained': False, 'model-path': "",
            'lidar-feat-net': {'name': "lidar-feat-pointseg", 'pretrained': False, 'model-path': "",
                               'requires-grad': True},
            'imu-feat-net': {'name': "imu-feat-rnn", 'pretrained': False, 'model-path': "",
                             'requires-grad': True},
            'odom-feat-net': {'name': "odom-feat-rnn", 'pretrained': False, 'model-path': "",
                              'requires-grad': True},
            'fusion-net': {'name': "fusion-layer-soft", 'requires-grad': True},
        },
        'lidar-feat-pointseg': {'dropout': 0.1, 'classes': ['unknown', 'object'], 'bypass': "simple",
                                'fusion': 'add', 'part': "encoder"},
        'lidar-feat-flownet': {'dropout': 0., 'fusion': 'add'},
        'lidar-feat-resnet': {'dropout': 0.25, 'fusion': 'add'},
        'lidar-feat-simple-1': {'dropout': 0.25, 'fusion': 'add', 'bypass': False},
        'imu-feat-fc': {'input-size': 6, 'hidden-size': [128, 256, 512, 512, 256, 128], 'dropout': 0.},
        'imu-feat-rnn': {'type': "lstm", 'input-size': 6, 'hidden-size': 128, 'num-layers': 2,
                         'bidirectional': True, 'dropout': 0.1},
        'fusion-layer-cat': {'type': "cat"},
        'fusion-layer-soft': {'type': "soft"},
        'odom-feat-fc': {'size': [1024, 512, 256], 'dropout': 0.},
        'odom-feat-rnn': {'type': "lstm", 'hidden-size': 1024, 'num-layers': 2, 'bidirectional': True,
                          'dropout': 0.},
        'losses': {'active': 'hwsloss',
                   'hwsloss': {'params': {'learn': True, 'sx': 0., 'sq': -3.}},
                   'lwsloss': {'params': {'beta': 1125.}},
                   'loss-type': "local+global"},
        'current-dataset': 'kitti',
        'channels': [0, 1, 2, 4, 5, 6],
        'optimizer': 'adam',
    }


def make_config(lidar="lidar-feat-pointseg", imu="imu-feat-rnn", fusion="fusion-layer-soft",
                odom="odom-feat-rnn", seq=2, overrides=None):
    """Convenience: default config with the four sub-net names and S = `seq` consecutive pairs."""
    cfg = default_config()
    cfg['datasets']['sequence-size'] = seq
    cfg['datasets']['combinations'] = [[i, i + 1] for i in range(seq)]
    cfg['deeplio']['lidar-feat-net']['name'] = lidar
    cfg['deeplio']['imu-feat-net']['name'] = imu
    cfg['deeplio']['fusion-net']['name'] = fusion
    cfg['deeplio']['odom-feat-net']['name'] = odom
    for path, val in (overrides or {}).items():
        node = cfg
        keys = path.split('/')
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = val
    return cfg


def load_config(path):
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


def clone(cfg):
    return copy.deepcopy(cfg)
