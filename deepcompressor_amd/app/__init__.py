"""Callers of the hot path that own its data formats: the per-layer weight / activation stages and the checkpoint
files they leave behind.  Model loading, datasets, evaluation and the CLI of the reference's ``app`` package are
not part of this package."""

from .llm import (ActivationEntry, LinearEntry, LlmActivationQuantizer, LlmActivationQuantizerConfig, LlmQuantizer, LlmQuantizerConfig,
                  LlmWeightQuantizer, LlmWeightQuantizerConfig, load_checkpoint, load_weight_checkpoint, prepare_layer_hessians,
                  quantize_layer_activations, quantize_layer_weights, save_checkpoint, save_weight_checkpoint)

from .llm_stages import (LlmQuantConfig, quantize_llm_layer_activations, quantize_llm_layer_weights, reorder_llm_layer,
                         rotate_llm, smooth_llm_layer)
from .llm_struct import (AttentionConfigStruct, FeedForwardConfigStruct, LlmFeedForwardStruct, LlmModelStruct,
                         LlmSelfAttentionStruct, LlmTransformerBlockStruct, LlmTransformerStruct)

__all__ = ["LlmQuantConfig", "rotate_llm", "reorder_llm_layer", "smooth_llm_layer", "quantize_llm_layer_weights",
           "quantize_llm_layer_activations", "AttentionConfigStruct", "FeedForwardConfigStruct",
           "LlmFeedForwardStruct", "LlmModelStruct", "LlmSelfAttentionStruct", "LlmTransformerBlockStruct",
           "LlmTransformerStruct", "ActivationEntry", "quantize_layer_activations", "LinearEntry", "LlmActivationQuantizer", "LlmActivationQuantizerConfig", "LlmQuantizer", "LlmQuantizerConfig",
           "LlmWeightQuantizer", "LlmWeightQuantizerConfig", "load_weight_checkpoint", "quantize_layer_weights",
           "save_weight_checkpoint", "save_checkpoint", "load_checkpoint", "prepare_layer_hessians"]
