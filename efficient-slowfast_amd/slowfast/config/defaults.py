"""Default config tree: every key of the reference's defaults (config/defaults.py:12-613) and of its
custom_config.py:9-34 additions, with the same default values, so any reference YAML / KEY VALUE override
list loads unchanged.  Only the keys read by the model constructors influence the hot path;
the rest exist so that unknown-key errors behave as in the reference."""
from .cfgnode import CfgNode

_DEFAULTS = {
    "BN": {"USE_PRECISE_STATS": False, "NUM_BATCHES_PRECISE": 200, "WEIGHT_DECAY": 0.0,
           "NORM_TYPE": "batchnorm", "NUM_SPLITS": 1, "NUM_SYNC_DEVICES": 1},
    "TRAIN": {"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 64, "EVAL_PERIOD": 1,
              "CHECKPOINT_PERIOD": 1, "AUTO_RESUME": True, "CHECKPOINT_FILE_PATH": "",
              "CHECKPOINT_TYPE": "pytorch", "CHECKPOINT_INFLATE": False, "TOPK": 5},
    "TEST": {"ENABLE": True, "DATASET": "kinetics", "BATCH_SIZE": 8, "CHECKPOINT_FILE_PATH": "",
             "NUM_ENSEMBLE_VIEWS": 10, "NUM_SPATIAL_CROPS": 3, "CHECKPOINT_TYPE": "pytorch"},
    "RESNET": {"TRANS_FUNC": "bottleneck_transform", "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64,
               "INPLACE_RELU": True, "STRIDE_1X1": False, "ZERO_INIT_FINAL_BN": False, "DEPTH": 50,
               "NUM_BLOCK_TEMP_KERNEL": [[3], [4], [6], [3]], "SPATIAL_STRIDES": [[1], [2], [2], [2]],
               "SPATIAL_DILATIONS": [[1], [1], [1], [1]]},
    "NONLOCAL": {"LOCATION": [[[]], [[]], [[]], [[]]], "GROUP": [[1], [1], [1], [1]],
                 "INSTANTIATION": "dot_product",
                 "POOL": [[[1, 2, 2], [1, 2, 2]], [[1, 2, 2], [1, 2, 2]], [[1, 2, 2], [1, 2, 2]],
                          [[1, 2, 2], [1, 2, 2]]]},
    "MODEL": {"ARCH": "slowfast", "MODEL_NAME": "SlowFast", "NUM_CLASSES": 400, "LOSS_FUNC": "cross_entropy",
              "SINGLE_PATHWAY_ARCH": ["c2d", "i3d", "slow", "fast"], "MULTI_PATHWAY_ARCH": ["slowfast"],
              "DROPOUT_RATE": 0.5, "FC_INIT_STD": 0.01, "HEAD_ACT": "softmax", "WEIGHTED_RANDOM_SAMPLER": False},
    "SLOWFAST": {"BETA_INV": 8, "ALPHA": 8, "FUSION_CONV_CHANNEL_RATIO": 2, "FUSION_KERNEL_SZ": 5,
                 "WIDTH_MULTI": 2.0, "GROUPS": 1},
    "DATA": {"PATH_TO_DATA_DIR": "/dataset/guojietian/kinetics400/", "PATH_LABEL_SEPARATOR": " ",
             "PATH_PREFIX": "", "CROP_SIZE": 224, "NUM_FRAMES": 8, "SAMPLING_RATE": 8,
             "MEAN": [0.45, 0.45, 0.45], "INPUT_CHANNEL_NUM": [3, 3], "STD": [0.225, 0.225, 0.225],
             "TRAIN_JITTER_SCALES": [256, 320], "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256, "TARGET_FPS": 30,
             "DECODING_BACKEND": "pyav", "INV_UNIFORM_SAMPLE": False, "RANDOM_FLIP": True, "MULTI_LABEL": False,
             "ENSEMBLE_METHOD": "sum", "REVERSE_INPUT_CHANNEL": False,
             "PATH_TO_TRAIN_DATA_TXT": "train_data_191105.txt",
             "PATH_TO_VAL_DATA_TXT": "train_data_for_191025_test.txt", "HALF_FACE": False},
    "SOLVER": {"BASE_LR": 0.1, "LR_POLICY": "cosine", "GAMMA": 0.1, "STEP_SIZE": 1, "STEPS": [], "LRS": [],
               "MAX_EPOCH": 300, "MOMENTUM": 0.9, "DAMPENING": 0.0, "NESTEROV": True, "WEIGHT_DECAY": 1e-4,
               "WARMUP_FACTOR": 0.1, "WARMUP_EPOCHS": 0.0, "WARMUP_START_LR": 0.01, "OPTIMIZING_METHOD": "sgd"},
    "NUM_GPUS": 1, "NUM_SHARDS": 1, "SHARD_ID": 0, "OUTPUT_DIR": "./tmp", "RNG_SEED": 1, "LOG_PERIOD": 10,
    "LOG_MODEL_INFO": True, "DIST_BACKEND": "nccl",
    "BENCHMARK": {"NUM_EPOCHS": 5, "LOG_PERIOD": 100, "SHUFFLE": True},
    "DATA_LOADER": {"NUM_WORKERS": 8, "PIN_MEMORY": True, "ENABLE_MULTI_THREAD_DECODE": False},
    "DETECTION": {"ENABLE": False, "ALIGNED": True, "SPATIAL_SCALE_FACTOR": 16, "ROI_XFORM_RESOLUTION": 7},
    "AVA": {"FRAME_DIR": "/mnt/fair-flash3-east/ava_trainval_frames.img/",
            "FRAME_LIST_DIR": "/mnt/vol/gfsai-flash3-east/ai-group/users/haoqifan/ava/frame_list/",
            "ANNOTATION_DIR": "/mnt/vol/gfsai-flash3-east/ai-group/users/haoqifan/ava/frame_list/",
            "TRAIN_LISTS": ["train.csv"], "TEST_LISTS": ["val.csv"],
            "TRAIN_GT_BOX_LISTS": ["ava_train_v2.2.csv"], "TRAIN_PREDICT_BOX_LISTS": [],
            "TEST_PREDICT_BOX_LISTS": ["ava_val_predicted_boxes.csv"], "DETECTION_SCORE_THRESH": 0.9,
            "BGR": False, "TRAIN_USE_COLOR_AUGMENTATION": False, "TRAIN_PCA_JITTER_ONLY": True,
            "TRAIN_PCA_EIGVAL": [0.225, 0.224, 0.229],
            "TRAIN_PCA_EIGVEC": [[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140],
                                 [-0.5836, -0.6948, 0.4203]],
            "TEST_FORCE_FLIP": False, "FULL_TEST_ON_VAL": False,
            "LABEL_MAP_FILE": "ava_action_list_v2.2_for_activitynet_2019.pbtxt",
            "EXCLUSION_FILE": "ava_val_excluded_timestamps_v2.2.csv", "GROUNDTRUTH_FILE": "ava_val_v2.2.csv",
            "IMG_PROC_BACKEND": "cv2"},
    "MULTIGRID": {"EPOCH_FACTOR": 1.5, "SHORT_CYCLE": False, "SHORT_CYCLE_FACTORS": [0.5, 0.5 ** 0.5],
                  "LONG_CYCLE": False,
                  "LONG_CYCLE_FACTORS": [(0.25, 0.5 ** 0.5), (0.5, 0.5 ** 0.5), (0.5, 1), (1, 1)],
                  "BN_BASE_SIZE": 8, "EVAL_FREQ": 3, "LONG_CYCLE_SAMPLING_RATE": 0, "DEFAULT_B": 0,
                  "DEFAULT_T": 0, "DEFAULT_S": 0},
    "TENSORBOARD": {"ENABLE": True, "LOG_DIR": "", "CLASS_NAMES_PATH": "", "CATEGORIES_PATH": "",
                    "CONFUSION_MATRIX": {"ENABLE": False, "FIGSIZE": [8, 8], "SUBSET_PATH": ""},
                    "HISTOGRAM": {"ENABLE": False, "SUBSET_PATH": "", "TOPK": 3, "FIGSIZE": [8, 8]},
                    "MODEL_VIS": {"ENABLE": False}},
    "DEMO": {"ENABLE": False, "LABEL_FILE_PATH": "", "DATA_SOURCE": "", "DISPLAY_WIDTH": 0, "DISPLAY_HEIGHT": 0,
             "DETECTRON2_OBJECT_DETECTION_MODEL_CFG": "", "DETECTRON2_OBJECT_DETECTION_MODEL_WEIGHTS": "",
             "OUTPUT_FILE": ""},
}

_C = CfgNode(_DEFAULTS)


def _assert_and_infer_cfg(cfg):
    """Same consistency checks as the reference (config/defaults.py:616-636)."""
    if cfg.BN.USE_PRECISE_STATS:
        assert cfg.BN.NUM_BATCHES_PRECISE >= 0
    assert cfg.TRAIN.CHECKPOINT_TYPE in ["pytorch", "caffe2"]
    assert cfg.TRAIN.BATCH_SIZE % cfg.NUM_GPUS == 0
    assert cfg.TEST.CHECKPOINT_TYPE in ["pytorch", "caffe2"]
    assert cfg.TEST.BATCH_SIZE % cfg.NUM_GPUS == 0
    assert cfg.TEST.NUM_SPATIAL_CROPS == 3
    assert cfg.RESNET.NUM_GROUPS > 0
    assert cfg.RESNET.WIDTH_PER_GROUP > 0
    assert cfg.RESNET.WIDTH_PER_GROUP % cfg.RESNET.NUM_GROUPS == 0
    assert cfg.SHARD_ID < cfg.NUM_SHARDS
    return cfg


def get_cfg():
    """A fresh copy of the defaults (config/defaults.py:639-643)."""
    return _assert_and_infer_cfg(_C.clone())
