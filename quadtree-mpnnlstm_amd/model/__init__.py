"""Drop-in mirror of the reference's `model` package (model/{graph_functions,model,seq2seq,
mpnnlstm,utils}.py) for the Quadtree-MPNNLSTM hot path, running on MI355X through
libqtmpnn_hip.so.  Put `quadtree-mpnnlstm_amd/` on sys.path and import exactly as with the
reference: `from model.mpnnlstm import NextFramePredictorS2S`."""
