"""Hidden sizes of the LLMs whose precomputed note embeddings feed the fusion blocks.

Mirrors the alias table of the reference's fusions/load_llm.py:5-13 as a local lookup: the reference asks the
HF hub (AutoConfig.from_pretrained, :30-31), which needs network access; the fusion constructors only need the
integer.  Raw-text mode (use_text_embeddings=False: tokenise + run a frozen LLM, :79-201) is outside the hot
path this package accelerates and is not provided.
"""

_D_MODEL = {
    "GPT2": 768, "GPT2M": 1024, "GPT2L": 1280, "GPT2XL": 1600, "BERT": 768, "Llama": 4096, "DeepSeek": 4096,
    "openai-community/gpt2": 768, "openai-community/gpt2-medium": 1024, "openai-community/gpt2-large": 1280,
    "openai-community/gpt2-xl": 1600, "google-bert/bert-base-uncased": 768, "meta-llama/Llama-3.1-8B": 4096,
    "deepseek-ai/deepseek-llm-7b-base": 4096,
}


def register_d_model(name: str, d_model: int) -> None:
    """Teach the table a new alias (tests use small toy widths)."""
    _D_MODEL[name] = int(d_model)


def get_d_model(llm_model_fusion: str) -> int:
    try:
        return _D_MODEL[llm_model_fusion]
    except KeyError:
        raise KeyError(
            f"unknown LLM alias {llm_model_fusion!r}: add it with fusions.load_llm.register_d_model(name, d_model)") from None


_CONTEXT = {"GPT2": 1024, "GPT2M": 1024, "GPT2L": 1024, "GPT2XL": 1024, "BERT": 512, "Llama": 131072, "DeepSeek": 4096,
            "openai-community/gpt2": 1024, "openai-community/gpt2-medium": 1024, "openai-community/gpt2-large": 1024,
            "openai-community/gpt2-xl": 1024, "google-bert/bert-base-uncased": 512, "meta-llama/Llama-3.1-8B": 131072,
            "deepseek-ai/deepseek-llm-7b-base": 4096}


def get_context_window_size(llm_model_fusion: str, device="cpu") -> int:
    """positional context window of the LLM (reference :38-76 loads the model to read it; a table here, same values
    as the alias comments at :5-13)"""
    try:
        return _CONTEXT[llm_model_fusion]
    except KeyError:
        raise KeyError(f"unknown LLM alias {llm_model_fusion!r}") from None


def load_llm(*_a, **_k):
    raise NotImplementedError("raw-text fusion (use_text_embeddings=False) is not part of the MI355X hot path; "
                              "precompute note embeddings and pass use_text_embeddings=True")


def embed_notes(*_a, **_k):
    raise NotImplementedError("raw-text fusion (use_text_embeddings=False) is not part of the MI355X hot path")


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
