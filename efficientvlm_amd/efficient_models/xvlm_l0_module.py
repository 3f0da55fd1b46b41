"""Hard-concrete L0 gate module — drop-in for the reference's efficient_models/xvlm_l0_module.py:XVLML0Module
(same constructor, attributes, parameter names `{vision,text,cross}_{head,int}_loga`, `lambda_1/2`, and methods
forward / lagrangian_regularization / constrain_parameters / set_lagrangian_warmup_steps / calculate_model_size).

Gate sampling (train) and the deterministic top-k masks (eval) are HIP kernels (evlm_l0_sample_*, evlm_l0_deterministic);
eval-mode masks are bit-exact against the reference (tests/test_ops_gpu.py).  The 37 k-element expected-sparsity
Lagrangian is scalar glue left to PyTorch.
"""
import math
import os

import numpy as np
import torch
from torch.nn.modules import Module
from torch.nn.parameter import Parameter

from .. import ops
from ..runtime import BertConfig, read_json

limit_a, limit_b, epsilon = -.1, 1.1, 1e-6


class XVLML0Module(Module):
    with_decoder = False       # generation_l0_module.VQAL0Module: + gates for the answer decoder's layers

    def __init__(self, config, droprate_init=0.5, temperature=2. / 3., lagrangian_warmup=0, start_sparsity=0.0,
                 target_sparsity=0.0, pruning_type="structured_heads+structured_mlp", magical_number=0.8):
        super().__init__()
        tc = config["text_encoder"]
        text_config = BertConfig.from_any(tc) if isinstance(tc, dict) else BertConfig.from_json_file(os.path.join(tc, "config.json"))
        text_config.num_hidden_layers = config["text_num_hidden_layers"] if "text_num_hidden_layers" in config else 12
        assert text_config.num_hidden_layers in [6, 12], "param initialization not implemented"
        text_config.fusion_layer = text_config.num_hidden_layers // 2
        vision_config = read_json(config["vision_config"])
        assert config["patch_size"] == vision_config["patch_size"]
        self.all_types = ["vision_intermediate_z", "vision_head_z", "text_intermediate_z", "text_head_z",
                          "cross_intermediate_z", "cross_head_z"] + (["decoder_head_z", "decoder_intermediate_z"]
                                                                     if self.with_decoder else [])
        self.pruning_type = pruning_type
        self.hidden_size = text_config.hidden_size
        self.intermediate_size = text_config.intermediate_size
        self.num_attention_heads = text_config.num_attention_heads
        self.dim_per_head = self.hidden_size // self.num_attention_heads
        self.vision_num_hidden_layers = vision_config["num_hidden_layers"]
        self.text_num_hidden_layers = text_config.fusion_layer
        self.cross_num_hidden_layers = text_config.num_hidden_layers - text_config.fusion_layer
        self.decoder_num_hidden_layers = self.cross_num_hidden_layers if self.with_decoder else 0   # generation_l0_module.py:47
        self.mlp_num_per_layer = 1
        self.params_per_head_layer = self.hidden_size * self.hidden_size * 4 + self.hidden_size * 4
        self.params_per_head = self.params_per_head_layer // self.num_attention_heads
        self.params_per_mlp_layer = self.hidden_size * self.intermediate_size * 2 + self.hidden_size + self.hidden_size * 4
        self.params_per_intermediate_dim = self.params_per_mlp_layer // self.intermediate_size
        self.full_model_size = (self.params_per_head_layer + self.params_per_mlp_layer) * self.vision_num_hidden_layers + \
                               (self.params_per_head_layer + self.params_per_mlp_layer) * self.text_num_hidden_layers + \
                               (self.params_per_head_layer * 2 + self.params_per_mlp_layer) * (self.cross_num_hidden_layers
                                                                                              + self.decoder_num_hidden_layers)
        self.prunable_model_size = 0
        self.temperature = temperature
        self.droprate_init = droprate_init if droprate_init != 0. else 0.5
        self.types, self.z_logas, self.parameters_per_dim, self.sizes, self.shapes = [], {}, {}, {}, {}
        types = self.pruning_type.split("+")
        for t in types:
            if t != "layer":
                self.initialize_one_module(t)
        self.magical_number = magical_number
        self.lambda_1 = torch.nn.Parameter(torch.tensor(0.0))
        self.lambda_2 = torch.nn.Parameter(torch.tensor(0.0))
        self.lagrangian_warmup = lagrangian_warmup
        self.start_sparsity = start_sparsity
        self.target_sparsity = target_sparsity
        self.injected_eps = None   # test hook: dict type -> uniform draws, used once instead of get_eps

    def set_lagrangian_warmup_steps(self, lagrangian_warmup):
        self.lagrangian_warmup = lagrangian_warmup

    def initialize_one_module(self, module_name):
        if module_name == "structured_heads":
            self.initialize_structured_head()
        elif module_name == "structured_mlp":
            self.initialize_structured_mlp()

    def add_one_module(self, z_loga, type, parameter_per_dim, size, shape):
        self.types.append(type)
        self.z_logas[type] = z_loga
        self.parameters_per_dim[type] = parameter_per_dim
        self.sizes[type] = size
        self.shapes[type] = shape

    def initialize_parameters(self, size, num_layer=None):
        return Parameter(torch.zeros(num_layer, size)) if num_layer is not None else Parameter(torch.zeros(size))

    def initialize_structured_head(self, add_prunable_model_size=True):
        H = self.num_attention_heads
        self.vision_head_loga = self.initialize_parameters(H, self.vision_num_hidden_layers)
        self.text_head_loga = self.initialize_parameters(H, self.text_num_hidden_layers)
        self.cross_head_loga = self.initialize_parameters(H, self.cross_num_hidden_layers * 2)
        if self.with_decoder:
            self.decoder_head_loga = self.initialize_parameters(H, self.decoder_num_hidden_layers * 2)
        for p in (self.vision_head_loga, self.text_head_loga, self.cross_head_loga) + \
                ((self.decoder_head_loga,) if self.with_decoder else ()):
            self.reset_loga(p, mean=10)
        self.add_one_module(self.vision_head_loga, "vision_head", self.params_per_head, H, [self.vision_num_hidden_layers, 1, H, 1, 1])
        self.add_one_module(self.text_head_loga, "text_head", self.params_per_head, H, [self.text_num_hidden_layers, 1, H, 1, 1])
        self.add_one_module(self.cross_head_loga, "cross_head", self.params_per_head, H, [self.cross_num_hidden_layers * 2, 1, H, 1, 1])
        if self.with_decoder:
            self.add_one_module(self.decoder_head_loga, "decoder_head", self.params_per_head, H,
                                [self.decoder_num_hidden_layers * 2, 1, H, 1, 1])
        if add_prunable_model_size:
            self.prunable_model_size += self.params_per_head * H * (self.vision_num_hidden_layers + self.text_num_hidden_layers
                                                                    + self.cross_num_hidden_layers * 2
                                                                    + self.decoder_num_hidden_layers * 2)

    def initialize_structured_mlp(self):
        f = self.intermediate_size
        self.vision_int_loga = self.initialize_parameters(f, self.vision_num_hidden_layers)
        self.text_int_loga = self.initialize_parameters(f, self.text_num_hidden_layers)
        self.cross_int_loga = self.initialize_parameters(f, self.cross_num_hidden_layers)
        self.add_one_module(self.vision_int_loga, "vision_intermediate", self.params_per_intermediate_dim, f, [self.vision_num_hidden_layers, 1, 1, f])
        self.add_one_module(self.text_int_loga, "text_intermediate", self.params_per_intermediate_dim, f, [self.text_num_hidden_layers, 1, 1, f])
        self.add_one_module(self.cross_int_loga, "cross_intermediate", self.params_per_intermediate_dim, f, [self.cross_num_hidden_layers, 1, 1, f])
        if self.with_decoder:
            self.decoder_int_loga = self.initialize_parameters(f, self.decoder_num_hidden_layers)
            self.add_one_module(self.decoder_int_loga, "decoder_intermediate", self.params_per_intermediate_dim, f,
                                [self.decoder_num_hidden_layers, 1, 1, f])
        self.prunable_model_size += self.params_per_mlp_layer * (self.vision_num_hidden_layers + self.text_num_hidden_layers
                                                                 + self.cross_num_hidden_layers + self.decoder_num_hidden_layers)
        for p in (self.vision_int_loga, self.text_int_loga, self.cross_int_loga) + \
                ((self.decoder_int_loga,) if self.with_decoder else ()):
            self.reset_loga(p)

    def reset_loga(self, tensor, mean=None):
        if mean is None:
            mean = math.log(1 - self.droprate_init) - math.log(self.droprate_init)
        tensor.data.normal_(mean, 1e-2)

    def constrain_parameters(self):
        for key in self.z_logas:
            self.z_logas[key].data.clamp_(min=math.log(1e-2), max=math.log(1e2))

    def cdf_qz(self, x, loga):
        xn = (x - limit_a) / (limit_b - limit_a)
        logits = math.log(xn) - math.log(1 - xn)
        return torch.sigmoid(logits * self.temperature - loga).clamp(min=epsilon, max=1 - epsilon)

    def get_num_parameters_and_constraint(self):
        n = 0
        for t in ("vision_head", "text_head", "cross_head", "vision_intermediate", "text_intermediate", "cross_intermediate") \
                + (("decoder_head", "decoder_intermediate") if self.with_decoder else ()):
            n = n + torch.sum(1 - self.cdf_qz(0, self.z_logas[t])) * self.parameters_per_dim[t]
        return n

    def get_target_sparsity(self, pruned_steps):
        if torch.is_tensor(pruned_steps):      # extension: a DEVICE step counter (a captured training step replays with a new one)
            ramp = torch.clamp(pruned_steps.to(torch.float32) / self.lagrangian_warmup, max=1.0)
            return (self.target_sparsity - self.start_sparsity) * ramp + self.start_sparsity
        return (self.target_sparsity - self.start_sparsity) * min(1, pruned_steps / self.lagrangian_warmup) + self.start_sparsity

    # extension: on the GPU the whole term - expected size over every gate type, ramped target, the two multiplier products -
    # is ONE kernel launch each way (ops.l0_lagrangian) instead of ~90 launches of a few microseconds; the tensor expressions
    # below remain for tensors that are not on the device.  EVLM_NO_FUSED_LAGRANGIAN=1: the expressions everywhere.
    fused_lagrangian = True

    def _lagrangian_types(self):
        return ("vision_head", "text_head", "cross_head", "vision_intermediate", "text_intermediate", "cross_intermediate") \
            + (("decoder_head", "decoder_intermediate") if self.with_decoder else ())

    def lagrangian_regularization(self, pruned_steps):
        import os
        if (self.fused_lagrangian and self.lambda_1.is_cuda and not os.environ.get("EVLM_NO_FUSED_LAGRANGIAN")
                and all(t in self.z_logas for t in self._lagrangian_types())):
            types = self._lagrangian_types()
            xn = (0 - limit_a) / (limit_b - limit_a)
            logit_c = (math.log(xn) - math.log(1 - xn)) * self.temperature
            return ops.l0_lagrangian([self.z_logas[t] for t in types], [self.parameters_per_dim[t] for t in types], logit_c,
                                     epsilon, self.prunable_model_size, self.target_sparsity, self.start_sparsity,
                                     self.lagrangian_warmup, pruned_steps, self.lambda_1, self.lambda_2)
        target_sparsity = self.target_sparsity
        expected_size = self.get_num_parameters_and_constraint()
        expected_sparsity = 1 - expected_size / self.prunable_model_size
        if self.lagrangian_warmup > 0:
            target_sparsity = self.get_target_sparsity(pruned_steps)
        lagrangian_loss = (self.lambda_1 * (expected_sparsity - target_sparsity)
                           + self.lambda_2 * (expected_sparsity - target_sparsity) ** 2)
        return lagrangian_loss, expected_sparsity, target_sparsity

    def get_eps(self, size):
        """uniform draws on the CPU generator, like the reference (xvlm_l0_module.py:239-244)"""
        return torch.empty(size).uniform_(epsilon, 1 - epsilon)

    # extension (captured training steps, trainer.ITRTrainer / VQATrainer): `static_eps` = {type: persistent device buffer} the
    # gate noise is READ from - the trainer refills the buffers before every replay with the draws this method would have
    # made (same generator, same order: `eps_trace` records (type, shape) of one eager forward);  None = draw here
    static_eps = None
    eps_trace = None

    def _sample_z(self, loga, type=None):
        if self.static_eps is not None and type in self.static_eps:
            return ops.l0_sample(loga, self.static_eps[type], self.temperature)
        if self.eps_trace is not None:
            self.eps_trace.append((type, tuple(loga.shape)))
        if self.injected_eps is not None and type in self.injected_eps:
            eps = self.injected_eps[type]
        else:
            eps = self.get_eps(tuple(loga.shape))
        return ops.l0_sample(loga, eps.to(loga.device), self.temperature)

    def _deterministic_z(self, size, loga):
        """one layer row; the batched kernel is used by forward()"""
        return ops.l0_deterministic(loga.view(1, -1), self.temperature, self.magical_number).view(-1)

    def get_z_from_zs(self, zs):
        out = {}
        for type in self.all_types:
            name = type[:-2]
            z = zs.get(type, np.ones(self.shapes[name]))
            out[name] = (z.squeeze().detach().cpu().numpy() > 0) if torch.is_tensor(z) else z
        return out

    def calculate_model_size(self, zs):
        nz = self.get_z_from_zs(zs)
        cnt = lambda k, layers, m: nz[k].reshape(layers, m).sum(-1).tolist()
        f, H = self.intermediate_size, self.num_attention_heads
        r_vi = cnt("vision_intermediate", self.vision_num_hidden_layers, f)
        r_ti = cnt("text_intermediate", self.text_num_hidden_layers, f)
        r_ci = cnt("cross_intermediate", self.cross_num_hidden_layers, f)
        r_vh = cnt("vision_head", self.vision_num_hidden_layers, H)
        r_th = cnt("text_head", self.text_num_hidden_layers, H)
        r_ch = cnt("cross_head", self.cross_num_hidden_layers * 2, H)
        head_nums = sum(r_ch) + sum(r_th) + sum(r_vh)
        intermediate_nums = sum(r_vi) + sum(r_ti) + sum(r_ci)
        extra = {}
        if self.with_decoder:                                   # generation_l0_module.py:307-338
            r_di = cnt("decoder_intermediate", self.decoder_num_hidden_layers, f)
            r_dh = cnt("decoder_head", self.decoder_num_hidden_layers * 2, H)
            head_nums += sum(r_dh)
            intermediate_nums += sum(r_di)
            extra = {"decoder_intermediate_dims": r_di, "decoder_head_nums": r_dh}
        remaining = head_nums * self.params_per_head + intermediate_nums * 2 * self.hidden_size
        pruned = self.prunable_model_size - remaining
        return {"vision_intermediate_dims": r_vi, "text_intermediate_dims": r_ti, "cross_intermediate_dims": r_ci,
                "vision_head_nums": r_vh, "text_head_nums": r_th, "cross_head_nums": r_ch, **extra, "pruned_params": pruned,
                "remaining_params": remaining, "pruned_model_sparsity": pruned / self.prunable_model_size}

    def forward(self, training=True):
        zs = {}
        for type in self.types:
            loga = self.z_logas[type]
            if training:
                zs[f"{type}_z"] = self._sample_z(loga, type).reshape(self.shapes[type])
            else:
                zs[f"{type}_z"] = ops.l0_deterministic(loga, self.temperature, self.magical_number).reshape(self.shapes[type])
        if training:
            self.injected_eps = None
        return zs
