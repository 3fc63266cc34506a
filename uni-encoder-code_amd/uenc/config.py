"""cfg keys of the reference's `model/config.py` (the config surface of boundary B1).

Same function names, same keys, same defaults — a YAML or `KEY VALUE` override written for the
reference configures this implementation unchanged.  Keys that only steer out-of-scope subsystems
(depth / pose / data augmentation / logging) are kept so the reference's YAML chain merges cleanly.
Reference: model/config.py:9-135 (common), :138-189 (uni_encoder), :192-214 (swin), :217-237 (dinat).
"""
from .d2 import CfgNode as CN

__all__ = ["add_common_config", "add_uni_encoder_config", "add_swin_config", "add_dinat_config",
           "add_convnext_config", "add_resnet_posenet_config"]


def _set(cfg, dotted: str, value):
    node = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        if p not in node:
            node[p] = CN()
        node = node[p]
    node[parts[-1]] = value


def _apply(cfg, table):
    for k, v in table:
        _set(cfg, k, v)


_COMMON = [
    ("INPUT.DATASET_MAPPER_NAME", "oneformer_unified"), ("INPUT.COLOR_AUG_SSD", False),
    ("INPUT.CROP.SINGLE_CATEGORY_MAX_AREA", 1.0), ("INPUT.SIZE_DIVISIBILITY", -1),
    ("INPUT.TASK_SEQ_LEN", 77), ("INPUT.MAX_SEQ_LEN", 77),
    ("INPUT.TASK_PROB.SEMANTIC", 0.33), ("INPUT.TASK_PROB.INSTANCE", 0.66),
    ("DATASETS.SEG_TEST_PANOPTIC", ("",)), ("DATASETS.SEG_TEST_INSTANCE", ("",)), ("DATASETS.SEG_TEST_SEMANTIC", ("",)),
    ("DATASETS.TRAIN", ("",)), ("DATASETS.DEPTH_TEST", ("",)),
    ("SOLVER.WEIGHT_DECAY_EMBED", 0.0), ("SOLVER.OPTIMIZER", "ADAMW"), ("SOLVER.BACKBONE_MULTIPLIER", 0.1),
    ("SOLVER.DISP_INIT_ITER", 0), ("SOLVER.MOTION_INIT_ITER", 10000), ("SOLVER.MASK_INIT_ITER", 20000),
    ("SOLVER.FINE_TUNE_ITER", 30000),
    ("WANDB.PROJECT", "OneFormer"), ("WANDB.NAME", None),
    ("MLFLOW.PROJECT", "MonoDepthTinyOneFormer"), ("MLFLOW.NAME", None), ("MLFLOW.TRACKING_URI", "http://localhost:5000"),
    ("MODEL.IS_TRAIN", True), ("MODEL.IS_DEMO", False),
    ("MODEL.TEXT_ENCODER.WIDTH", 256), ("MODEL.TEXT_ENCODER.CONTEXT_LENGTH", 77), ("MODEL.TEXT_ENCODER.NUM_LAYERS", 12),
    ("MODEL.TEXT_ENCODER.VOCAB_SIZE", 49408), ("MODEL.TEXT_ENCODER.PROJ_NUM_LAYERS", 2), ("MODEL.TEXT_ENCODER.N_CTX", 16),
    ("MODEL.TEST.SEMANTIC_ON", True), ("MODEL.TEST.INSTANCE_ON", False), ("MODEL.TEST.PANOPTIC_ON", False),
    ("MODEL.TEST.DEPTH_ON", False), ("MODEL.TEST.DETECTION_ON", False), ("MODEL.TEST.OBJECT_MASK_THRESHOLD", 0.0),
    ("MODEL.TEST.OVERLAP_THRESHOLD", 0.0), ("MODEL.TEST.SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE", False),
    ("MODEL.TEST.TASK", "panoptic"),
    ("TEST.AUG.IS_SLIDE", False), ("TEST.AUG.CROP_SIZE", (640, 640)), ("TEST.AUG.STRIDE", (426, 426)),
    ("TEST.AUG.SCALE", (2048, 640)), ("TEST.AUG.SETR_MULTI_SCALE", True), ("TEST.AUG.KEEP_RATIO", True),
    ("TEST.AUG.SIZE_DIVISOR", 32),
    ("MODEL.SEM_SEG_HEAD.MASK_DIM", 256), ("MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", 0),
    ("MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME", "BasePixelDecoder"), ("MODEL.SEM_SEG_HEAD.DEPTH_DECODER_NAME", "BasePixelDecoder"),
    ("MODEL.SEM_SEG_HEAD.SEM_EMBED_DIM", 256), ("MODEL.SEM_SEG_HEAD.INST_EMBED_DIM", 256),
    ("INPUT.IMAGE_SIZE", 1024), ("INPUT.MIN_SCALE", 0.1), ("INPUT.MAX_SCALE", 2.0),
    ("INPUT.SEG_MIN_SIZE_TRAIN", (192,)), ("INPUT.SEG_MAX_SIZE_TRAIN", 512), ("INPUT.SEG_MIN_SIZE_TRAIN_SAMPLING", "choice"),
    ("INPUT.SEG_MIN_SIZE_TEST", 192), ("INPUT.SEG_MAX_SIZE_TEST", 192),
    ("INPUT.DEPTH_MIN_SIZE_TRAIN", (192,)), ("INPUT.DEPTH_MAX_SIZE_TRAIN", 512),
    ("INPUT.DEPTH_MIN_SIZE_TRAIN_SAMPLING", "choice"), ("INPUT.DEPTH_MIN_SIZE_TEST", 192), ("INPUT.DEPTH_MAX_SIZE_TEST", 192),
    ("INPUT.SEG_CROP.ENABLED", False), ("INPUT.SEG_CROP.TYPE", "absolute"), ("INPUT.SEG_CROP.SIZE", (192, 512)),
    ("INPUT.SEG_CROP.SINGLE_CATEGORY_MAX_AREA", 1.0),
    ("INPUT.DEPTH_CROP.ENABLED", False), ("INPUT.DEPTH_CROP.TYPE", "absolute"), ("INPUT.DEPTH_CROP.SIZE", (192, 512)),
    ("INPUT.SEG_COLOR_AUG_SSD", False), ("INPUT.DEPTH_COLOR_JITTER", False),
    ("MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES", ["res3", "res4", "res5"]),
    ("MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_N_POINTS", 4),
    ("MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_N_HEADS", 8),
]

_UNI = [
    ("MODEL.ONE_FORMER.DEEP_SUPERVISION", True), ("MODEL.ONE_FORMER.NO_OBJECT_WEIGHT", 0.1),
    ("MODEL.ONE_FORMER.CLASS_WEIGHT", 1.0), ("MODEL.ONE_FORMER.DICE_WEIGHT", 1.0), ("MODEL.ONE_FORMER.MASK_WEIGHT", 20.0),
    ("MODEL.ONE_FORMER.CONTRASTIVE_WEIGHT", 0.5), ("MODEL.ONE_FORMER.MONODEPTH_WEIGHT", 2.0),
    ("MODEL.ONE_FORMER.OPTICAL_FLOW_DISTIL_WEIGHT", 1.0), ("MODEL.ONE_FORMER.CONTRASTIVE_TEMPERATURE", 0.07),
    ("MODEL.ONE_FORMER.NHEADS", 8), ("MODEL.ONE_FORMER.DROPOUT", 0.1), ("MODEL.ONE_FORMER.DIM_FEEDFORWARD", 2048),
    ("MODEL.ONE_FORMER.ENC_LAYERS", 0), ("MODEL.ONE_FORMER.CLASS_DEC_LAYERS", 2), ("MODEL.ONE_FORMER.DEC_LAYERS", 6),
    ("MODEL.ONE_FORMER.PRE_NORM", False), ("MODEL.ONE_FORMER.HIDDEN_DIM", 256),
    ("MODEL.ONE_FORMER.NUM_OBJECT_QUERIES", 120), ("MODEL.ONE_FORMER.NUM_OBJECT_CTX", 16),
    ("MODEL.ONE_FORMER.USE_TASK_NORM", True), ("MODEL.ONE_FORMER.TRANSFORMER_IN_FEATURE", "res5"),
    ("MODEL.ONE_FORMER.ENFORCE_INPUT_PROJ", False), ("MODEL.ONE_FORMER.SIZE_DIVISIBILITY", 32),
    ("MODEL.ONE_FORMER.TRANSFORMER_DECODER_NAME", "ContrastiveMultiScaleMaskedTransformerDecoder"),
    ("MODEL.ONE_FORMER.TRAIN_NUM_POINTS", 112 * 112), ("MODEL.ONE_FORMER.OVERSAMPLE_RATIO", 3.0),
    ("MODEL.ONE_FORMER.IMPORTANCE_SAMPLE_RATIO", 0.75),
]

_SWIN = [
    ("MODEL.SWIN.PRETRAIN_IMG_SIZE", 224), ("MODEL.SWIN.PATCH_SIZE", 4), ("MODEL.SWIN.EMBED_DIM", 96),
    ("MODEL.SWIN.DEPTHS", [2, 2, 6, 2]), ("MODEL.SWIN.NUM_HEADS", [3, 6, 12, 24]), ("MODEL.SWIN.WINDOW_SIZE", 7),
    ("MODEL.SWIN.MLP_RATIO", 4.0), ("MODEL.SWIN.QKV_BIAS", True), ("MODEL.SWIN.QK_SCALE", None),
    ("MODEL.SWIN.DROP_RATE", 0.0), ("MODEL.SWIN.ATTN_DROP_RATE", 0.0), ("MODEL.SWIN.DROP_PATH_RATE", 0.3),
    ("MODEL.SWIN.APE", False), ("MODEL.SWIN.PATCH_NORM", True),
    ("MODEL.SWIN.OUT_FEATURES", ["res2", "res3", "res4", "res5"]), ("MODEL.SWIN.USE_CHECKPOINT", False),
]

_DINAT = [
    ("MODEL.DiNAT.DEPTHS", [3, 4, 18, 5]), ("MODEL.DiNAT.OUT_FEATURES", ["res2", "res3", "res4", "res5"]),
    ("MODEL.DiNAT.EMBED_DIM", 64), ("MODEL.DiNAT.MLP_RATIO", 3.0), ("MODEL.DiNAT.NUM_HEADS", [2, 4, 8, 16]),
    ("MODEL.DiNAT.DROP_PATH_RATE", 0.2), ("MODEL.DiNAT.KERNEL_SIZE", 7),
    ("MODEL.DiNAT.DILATIONS", [[1, 16, 1], [1, 4, 1, 8], [1, 2, 1, 3, 1, 4], [1, 2, 1, 2, 1]]),
    ("MODEL.DiNAT.OUT_INDICES", (0, 1, 2, 3)), ("MODEL.DiNAT.QKV_BIAS", True), ("MODEL.DiNAT.QK_SCALE", None),
    ("MODEL.DiNAT.DROP_RATE", 0), ("MODEL.DiNAT.ATTN_DROP_RATE", 0.0), ("MODEL.DiNAT.IN_PATCH_SIZE", 4),
]


def add_common_config(cfg):
    _apply(cfg, _COMMON)


def add_uni_encoder_config(cfg):
    _apply(cfg, _UNI)


def add_swin_config(cfg):
    _apply(cfg, _SWIN)


def add_dinat_config(cfg):
    _apply(cfg, _DINAT)


def add_convnext_config(cfg):
    """ConvNeXt backbone keys (model/config.py: add_convnext_config); the backbone itself is out of scope."""
    _apply(cfg, [("MODEL.CONVNEXT.IN_CHANNELS", 3), ("MODEL.CONVNEXT.DEPTHS", [3, 3, 27, 3]),
                 ("MODEL.CONVNEXT.DIMS", [192, 384, 768, 1536]), ("MODEL.CONVNEXT.DROP_PATH_RATE", 0.4),
                 ("MODEL.CONVNEXT.LSIT", 1.0), ("MODEL.CONVNEXT.OUT_INDICES", [0, 1, 2, 3]),
                 ("MODEL.CONVNEXT.OUT_FEATURES", ["res2", "res3", "res4", "res5"])])


def add_resnet_posenet_config(cfg):
    """Pose-net keys (model/config.py: add_resnet_posenet_config; sequence branch, out of scope) so shared YAMLs merge."""
    _apply(cfg, [("MODEL.POSE_RESNETS.NORM", "SyncBN"), ("MODEL.POSE_RESNETS.STEM_OUT_CHANNELS", 64),
                 ("MODEL.POSE_RESNETS.OUT_FEATURES", ["res5"]), ("MODEL.POSE_RESNETS.DEPTH", 18),
                 ("MODEL.POSE_RESNETS.NUM_GROUPS", 1), ("MODEL.POSE_RESNETS.WIDTH_PER_GROUP", 64),
                 ("MODEL.POSE_RESNETS.RES2_OUT_CHANNELS", 64), ("MODEL.POSE_RESNETS.STRIDE_IN_1X1", False),
                 ("MODEL.POSE_RESNETS.RES5_DILATION", 1),
                 ("MODEL.POSE_RESNETS.DEFORM_ON_PER_STAGE", [False, False, False, False]),
                 ("MODEL.POSE_RESNETS.DEFORM_MODULATED", False), ("MODEL.POSE_RESNETS.DEFORM_NUM_GROUPS", 1)])
