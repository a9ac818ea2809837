"""The configuration keys the hot path reads, with the reference's defaults after
`cfg_from_file('cfgs/res50.yml')` (lib/model/utils/config.py:19-310, cfgs/res50.yml).

A plain attribute namespace (the reference uses a global EasyDict `cfg`; same access syntax:
cfg.TRAIN.BATCH_SIZE, cfg['TEST'].RPN_NMS_THRESH ...).
"""


class _NS(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def default_cfg():
    c = _NS()
    c.TRAIN = _NS(
        MARGIN=-0.3,                         # config.py:23
        TRUNCATED=False,
        BATCH_SIZE=128,                      # RoIs sampled per image = P in training (config.py:81)
        FG_FRACTION=0.25, FG_THRESH=0.5, BG_THRESH_HI=0.5,
        BG_THRESH_LO=0.0,                    # cfgs/res50.yml overrides 0.1 -> 0.0
        BBOX_NORMALIZE_TARGETS_PRECOMPUTED=True,
        BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0), BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2),
        BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0),
        RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3, RPN_CLOBBER_POSITIVES=False,
        RPN_FG_FRACTION=0.5, RPN_BATCHSIZE=256, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=12000,
        RPN_POST_NMS_TOP_N=2000, RPN_MIN_SIZE=8, RPN_BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0),
        RPN_POSITIVE_WEIGHT=-1.0, query_size=128, SCALES=(600,), MAX_SIZE=1000,
    )
    c.TEST = _NS(NMS=0.3, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300,
                 RPN_MIN_SIZE=16, SCALES=(600,), MAX_SIZE=1000)
    c.POOLING_MODE = 'align'                 # cfgs/res50.yml:17
    c.POOLING_SIZE = 7
    c.MAX_NUM_GT_BOXES = 20                  # 20 VOC / 50 COCO (trainval_net_voc.py:196-204)
    c.ANCHOR_SCALES = [8, 16, 32]            # VOC; COCO: [4, 8, 16, 32]
    c.ANCHOR_RATIOS = [0.5, 1, 2]
    c.FEAT_STRIDE = [16]
    c.RNG_SEED = 3
    return c


cfg = default_cfg()


def cfg_from_list(pairs):
    """cfg_from_list(['TRAIN.BATCH_SIZE', 300, 'ANCHOR_SCALES', [4, 8, 16, 32]])
    (same calling convention as lib/model/utils/config.py:392-408)."""
    assert len(pairs) % 2 == 0
    for k, v in zip(pairs[0::2], pairs[1::2]):
        d = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            d = d[p]
        if parts[-1] not in d:
            raise KeyError(k)
        d[parts[-1]] = v
