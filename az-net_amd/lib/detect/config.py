"""AZ-detect config, host-side mirror of the reference's `detect.config`
(lib/detect/config.py): a module-global `cfg` tree, YAML overrides that must name
existing keys with matching types, and the helpers tools/prop_az.py calls
(`cfg_from_file`, `cfg_set_path`, `cfg_load_thresh`, `cfg_set_mode`,
`get_output_dir`).  Keys and defaults are the reference's (file:line cited per
group); the proposal path itself only reads the subset listed in SURVEY.md section 5.
"""
import os
import os.path as osp
import pickle

import numpy as np


class edict(dict):
    """Attribute-style nested dict (the reference uses the `easydict` package)."""

    def __init__(self, d=None, **kw):
        dict.__init__(self)
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, edict):
            v = edict(v)
        dict.__setitem__(self, k, v)
        dict.__setattr__(self, k, v)

    __setitem__ = __setattr__

    def has_key(self, k):
        return k in self

    def iteritems(self):
        return self.items()


def _defaults():
    """Default option tree.  Values are the reference's (lib/detect/config.py:39-219)."""
    train = dict(        # config.py:39-104 -- unused by the proposal path; present so the
        SCALES=(600,),   # reference's YAML files (experiments/cfgs/*.yml) merge cleanly
        MAX_SIZE=1000, IMS_PER_BATCH=2, BATCH_SIZE=128, FG_FRACTION=0.25, AZ_POS_FRACTION=0.5,
        FG_THRESH=0.5, BG_THRESH_HI=0.5, BG_THRESH_LO=0.1, USE_FLIPPED=True, BBOX_REG=True,
        BBOX_THRESH=0.5, SNAPSHOT_ITERS=10000, USE_CACHE=False, SNAPSHOT_INFIX='',
        USE_PREFETCH=False, UN_NORMALIZE=False, NUM_PROPOSALS=2000, ANCHORS_PER_IMG=20,
        ADDREGIONS=[[0, 0, 1, 1], [0, 0, 0.8, 0.8], [0, 0.2, 0.8, 1], [0.2, 0, 1, 0.8],
                    [0.2, 0.2, 1, 1]])
    test = dict(         # config.py:109-133
        SCALES=(600,),       # short-side target of the single test scale
        MAX_SIZE=1000,       # long-side cap (experiments/cfgs/voc.yml:13 lowers it to 800)
        NMS=0.5,             # apply_nms threshold (post-detection only, test.py:467-484)
        SVM=False, BBOX_REG=True, DISPLAY=False,
        NUM_PROPOSALS=300,
        PREFETCH=2,          # (not in the reference) images test_proposals reads ahead in a worker thread; 0: none
        BATCH_IMAGES=1)      # (not in the reference) > 1: test_proposals searches up to that many consecutive images of one shape
                             # in lockstep (az_batch_launch): same boxes per image, every level's rois of the batch in one head pass
    # the 11 adjacency templates, relative to a region (config.py:149-154)
    subregion = [[0, 0, 1, 1],
                 [-0.5, 0, 0.5, 1], [0.5, 0, 1.5, 1], [0, -0.5, 1, 0.5], [0, 0.5, 1, 1.5],
                 [0, 0, 0.5, 1], [0.5, 0, 1, 1], [0, 0, 1, 0.5], [0, 0.5, 1, 1],
                 [0.25, 0, 0.75, 1], [0, 0.25, 1, 0.75]]
    append_temp = np.transpose(np.array([[[0, 0, 1, 1], [-0.25, 0, 1, 1], [0, 0, 1.25, 1],
                                          [0, -0.25, 1, 1], [0, 0, 1, 1.25],
                                          [-0.125, -0.125, 1.125, 1.125],
                                          [0.125, 0.125, 0.875, 0.875]]]), axes=[0, 2, 1])
    sear = dict(         # config.py:139-195
        SUBREGION=subregion, NUM_SUBREG=len(subregion),
        ZOOM_ERR_PROB=0.3, TRAIN_REP=8, ADJ_THRESH=0.1, EMB_OBJ_THRESH=0.5, EMB_REG_THRESH=0.25,
        SCALE_ADJ_CONF=False,
        Tc=0.05,                     # score threshold when FIXED_PROPOSAL_NUM is off
        FIXED_PROPOSAL_NUM=True,
        APPEND_BOXES=False, APPEND_TEMP=append_temp,
        MIN_SIDE=10,                 # px; smallest side considered for prediction and zoom
        BATCH_SIZE=10000,            # regions per forward chunk (voc.yml:17 uses 1000)
        AZ_CONV=['conv5_3'], FRCNN_CONV=['conv5_3'])
    return dict(
        TRAIN=train, TEST=test, SEAR=sear,
        DEDUP_BOXES=1. / 16.,        # feature-space dedup scale (config.py:206)
        PIXEL_MEANS=np.array([[[102.9801, 115.9465, 122.7717]]]),   # BGR (config.py:210)
        RNG_SEED=3, EPS=1e-14,
        ROOT_DIR=osp.abspath(osp.join(osp.dirname(__file__), '..', '..')),
        EXP_DIR='default')


cfg = edict(_defaults())
__C = cfg


def get_output_dir(imdb, net):
    """<ROOT>/output/<EXP_DIR>/<imdb.name>[/<net.name>] (config.py:221-231)."""
    path = osp.abspath(osp.join(__C.ROOT_DIR, 'output', __C.EXP_DIR, imdb.name))
    if net is None:
        return path
    return osp.join(path, net.name)


def _merge_a_into_b(a, b):
    """Clobber b's options with a's; a may only name keys b has, with the same type
    (config.py:233-262)."""
    if type(a) is not edict:
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        if k == 'PIXEL_MEANS':
            v = np.array(v)
        if type(b[k]) is not type(v):
            # YAML has no tuple literal; the reference's own files only override scalars,
            # lists and nested dicts
            if isinstance(b[k], tuple) and isinstance(v, list):
                v = tuple(v)
            else:
                raise ValueError(('Type mismatch ({} vs. {}) for config key: {}')
                                 .format(type(b[k]), type(v), k))
        if type(v) is edict:
            try:
                _merge_a_into_b(a[k], b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a YAML file and merge it into the defaults (config.py:264-270)."""
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = edict(yaml.safe_load(f))
    _merge_a_into_b(yaml_cfg, __C)


def cfg_set_mode(mode, thresh=None):
    """Train: Tz = 0, TRAIN.NUM_PROPOSALS; Test: Tz = thresh, TEST.NUM_PROPOSALS
    (config.py:272-280)."""
    if mode == 'Train':
        __C.SEAR.Tz = 0.0
        __C.SEAR.NUM_PROPOSALS = __C.TRAIN.NUM_PROPOSALS
    elif mode == 'Test':
        assert (thresh is not None), 'testing Tz is not set!'
        __C.SEAR.Tz = thresh
        __C.SEAR.NUM_PROPOSALS = __C.TEST.NUM_PROPOSALS


def cfg_load_thresh(filename):
    """thresh.pkl written by the tuner (config.py:282-287)."""
    with open(filename, 'rb') as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            return pickle.load(f, encoding='latin1')      # Python-2 pickles


def cfg_set_path(exp_dir):
    """config.py:289-295."""
    __C.EXP_DIR = 'default' if exp_dir is None else exp_dir
