from bayesian_torch_amd.models.dnn_to_bnn import dnn_to_bnn, get_kl_loss, bnn_linear_layer, bnn_conv_layer, bnn_lstm_layer  # noqa: F401
