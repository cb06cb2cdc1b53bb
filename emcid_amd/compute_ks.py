"""Keys of one layer for a request list (reference: emcid/compute_ks.py:21-41 ``compute_ks_text_encoder``)."""
from typing import Dict, List

from .compute_z import get_module_input_output_at_words


def compute_ks_text_encoder(model, tok, requests: List[Dict], hparams, layer: int):
    """(num_requests, d): mean fc2-input at the last subject token of each request's prompts."""
    layername = hparams.rewrite_module_tmp.format(layer)
    return get_module_input_output_at_words(model, tok, requests, layername,
                                            num_fact_token=hparams.num_edit_tokens)[0]
