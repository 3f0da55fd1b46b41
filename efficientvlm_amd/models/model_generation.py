"""VQA teacher — drop-in for models/model_generation.py:XVLMForVQA (reference :228-377, `train=True` branch): the same
network as the student without gates (efficient_models/model_generation.py holds the shared forward)."""
from ..efficient_models.model_generation import _VQABase, remap_vqa_checkpoint  # noqa: F401


class XVLMForVQA(_VQABase):
    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False, config_text=None)
        self._build(config)

    def forward(self, image, quesiton, answer=None, k=None, weights=None, train=True, output_attentions=None,
                output_hidden_states=None):
        if not train:
            raise NotImplementedError("answer ranking (rank_answer, models/model_generation.py:385-442) is evaluation code "
                                      "outside the distillation training path")
        return self._train_forward(image, quesiton, answer, k, weights, None, output_attentions, output_hidden_states)
