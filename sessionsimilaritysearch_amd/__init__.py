"""MI355X-native session-similarity retrieval hot path.

Drop-in surface of the reference's path:
  normalize, build_index, FlatIndex (f32 / bf16)  (index.py     <- test_amazon_filterd.py / util_amazon_filtered.py)
  BinaryFlatIndex, pack_sign_bits                 (index.py     <- fine_tune_ours.py IndexBinaryFlat branch)
  SessionEncoder (+ prepare_actions)              (encoder.py   <- model/model.py UnifyPoolingGraphLevelEncoder,
                                                                  util_amazon_filtered.sequence_to_graph)
  get_prediction_by_knn, get_p_r, SessionItems    (retrieval.py <- test_amazon_filterd.py:59-85)
  HeteroSAGE, GraphPooling, AttentionPooling, ... (variants.py  <- model/gnn.py, model/model.py variants)
  ShardedFlatIndex                                (distributed.py: corpus row-sharded over RCCL)
  SessionBatch, build_batch, synthetic_actions    (sessions.py  <- sequence_to_graph + Batch.from_data_list, host side)
"""
from ._lib import SssError, build, exported_symbols, lib  # noqa: F401

__all__ = ["SssError", "build", "exported_symbols", "lib"]
