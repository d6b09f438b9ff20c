"""Configuration object for the RMem hot path.

Mirrors the attribute names the reference reads on its config bag
(/root/reference/aot_plus/configs/models/default.py:3-27,
 configs/models/default_deaot.py:4-17, configs/models/r50_deaotl.py:4-40,
 configs/models/r50_aotl.py:4-40, configs/pre_vost.py:12-17) so that code written
against the reference's ``cfg`` works unchanged.  Only attributes consumed on the
inference hot path are carried ("config / flags").
"""
from __future__ import annotations


class ModelConfig:
    """Plain attribute bag, same spelling as the reference's cfg."""

    def __init__(self, model: str = "r50_deaotl", former_mem_len: int = 1,
                 latter_mem_len: int = 3):
        model = model.lower()
        # --- configs/models/default.py:3-27
        self.MODEL_ALIGN_CORNERS = True
        self.MODEL_ENCODER_EMBEDDING_DIM = 256
        self.MODEL_DECODER_INTERMEDIATE_LSTT = True
        self.MODEL_LINEAR_Q = True
        self.MODEL_NORM_INP = True
        self.MODEL_FREEZE_BN = True
        self.MODEL_MAX_OBJ_NUM = 10
        self.MODEL_IGNORE_TOKEN = True
        self.MODEL_SELF_HEADS = 8
        self.MODEL_ATT_HEADS = 8
        self.MODEL_EPSILON = 1e-5
        self.MODEL_LSTT_NUM = 3
        # --- RMem attributes (configs/models/r50_deaotl.py:7-28)
        self.FORMER_MEM_LEN = int(former_mem_len)
        self.LATTER_MEM_LEN = int(latter_mem_len)
        self.GRU_MEMORY = False
        self.TIME_ENCODE = False
        self.TIME_ENCODE_NORM = False
        self.USE_TEMPORAL_POSITIONAL_EMBEDDING = True
        self.TEMPORAL_POSITIONAL_EMBEDDING_SLOT_4 = True
        self.USE_MASK = False
        self.NO_LONG_MEMORY = False
        self.NO_MEMORY_GAP = False
        self.REVERSE_INFER = False
        self.TEST_LONG_TERM_MEM_GAP = 5
        # --- stage pre_vost overrides (configs/pre_vost.py:16-17)
        self.MODEL_LINEAR_Q = False
        self.MODEL_IGNORE_TOKEN = True

        if model == "r50_deaotl":
            # configs/models/default_deaot.py:9-15, r50_deaotl.py:32-36
            self.MODEL_NAME = "R50_DeAOTL_Temp_pe_Slot_4"
            self.MODEL_VOS = "deaot"
            self.MODEL_ENGINE = "deaotengine"
            self.MODEL_DECODER_INTERMEDIATE_LSTT = False
            self.MODEL_SELF_HEADS = 1
            self.MODEL_ATT_HEADS = 1
            self.MODEL_ENCODER = "resnet50"
            self.MODEL_ENCODER_DIM = [256, 512, 1024, 1024]
        elif model == "r50_aotl":
            self.MODEL_NAME = "R50_AOTL_Temp_pe_Slot_4"
            self.MODEL_VOS = "aot"
            self.MODEL_ENGINE = "aotengine"
            self.MODEL_ENCODER = "resnet50"
            self.MODEL_ENCODER_DIM = [256, 512, 1024, 1024]
        elif model == "swinb_aotl":
            # configs/models/swinb_aotl.py:7-16 + the RMem attributes of r50_aotl.py:7-28: the
            # shipped swinb config lacks them and AOT.__init__ raises
            self.MODEL_NAME = "SwinB_AOTL_Temp_pe_Slot_4"
            self.MODEL_VOS = "aot"
            self.MODEL_ENGINE = "aotengine"
            self.MODEL_ENCODER = "swin_base"
            self.MODEL_ALIGN_CORNERS = False
            self.MODEL_ENCODER_DIM = [128, 256, 512, 512]
        else:
            raise NotImplementedError(
                f"model config '{model}' is not part of the hot-path scope")

    @property
    def mem_cap(self) -> int:
        """K = FORMER_MEM_LEN + LATTER_MEM_LEN (transformer.py:973)."""
        return self.FORMER_MEM_LEN + self.LATTER_MEM_LEN


def get_config(model: str = "r50_deaotl", former_mem_len: int = 1,
               latter_mem_len: int = 3) -> ModelConfig:
    """Analogue of tools/get_config.py:4-6 + tools/eval.py:91-92,134-135."""
    return ModelConfig(model, former_mem_len, latter_mem_len)
