"""Drop-in alias: ``import samgraph.torch as sam`` (what the reference's example scripts do,
example/samgraph/train_graphsage.py) resolves to the MI355X-native implementation in xgnn_amd."""
