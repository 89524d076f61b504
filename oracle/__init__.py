"""CPU oracle for the session-similarity hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the checker (or as
the timed CPU baseline), never as the thing shipped.  The product package
``sessionsimilaritysearch_amd`` never imports this package and fails loudly when its HIP
library is missing.

What it restates (reference file:line -> oracle function):

* ``util_amazon_filtered.py:98-230`` (structure of ``sequence_to_graph``) + PyG
  ``Batch.from_data_list``          -> ``graph_ref``
* ``model/NodeEmbedding.py:128-138`` (``NodeAsinEmbedding``)   -> ``gnn_ref.embedding_lookup``
* ``model/gnn.py:43-81`` (``HeteroGGNN``) with PyG 2.0.4 ``GATConv`` / ``GatedGraphConv`` /
  ``HeteroConv`` semantics       -> ``gnn_ref.hetero_ggnn``
* ``model/gnn.py:183-217`` (``PositionalAttentionPooling``)    -> ``gnn_ref.pos_att_pool``
* ``model/model.py:279-351`` (encoder wiring)                 -> ``gnn_ref.encoder_forward``
* ``util_amazon_filtered.py:28-31`` (``normalize``)            -> ``search_ref.normalize``
* ``test_amazon_filterd.py:207-223,578`` (``build_index`` / ``index.search``) with faiss
  ``IndexFlatIP`` semantics (Appendix A.5)                    -> ``search_ref``
* ``test_amazon_filterd.py:59-85`` (neighbour item vote, p/r) -> ``search_ref``

PARITY UNPINNED: the reference ships no tests, golden vectors or
fixtures for this path, and its arithmetic lives in ``torch_geometric==2.0.4`` and ``faiss``
(``dependency.txt:2-3``), neither of which is installed here or installable (no network).
The restatement therefore follows the published algorithms of those libraries as written
down in SURVEY.md Appendix A.  What *is* pinned:

* ``normalize(np.ones(4)) == [0.5, 0.5, 0.5, 0.5]`` -- the one known-answer value in the
  reference (``test_amazon_filterd.py:866``);
* ``torch.nn.GRUCell`` / ``nn.Embedding`` / ``nn.Linear`` / ``F.leaky_relu`` from the
  installed torch are used directly inside the restatement;
* ``NodeAsinEmbedding`` from the reference's own ``model/NodeEmbedding.py`` (loaded by file
  path in the build container only) generated ``tests/golden/node_asin_embedding.npz``
  (script: ``tests/golden/make_golden.py``).
"""
