"""``LSTMFlipout`` -- drop-in for reference ``layers/flipout_layers/rnn_flipout.py:45-153``: an LSTM cell unrolled over
time whose input-to-hidden and hidden-to-hidden maps are two ``LinearFlipout`` layers (a fresh weight draw per time
step and per map, as in the reference); both run on the fused linear kernel, the gate arithmetic is plain torch."""
import torch

from ..base_variational_layer import BaseVariationalLayer_
from .linear_flipout import LinearFlipout

__all__ = ["LSTMFlipout"]


class LSTMFlipout(BaseVariationalLayer_):
    def __init__(self, in_features, out_features, prior_mean=0, prior_variance=1, posterior_mu_init=0,
                 posterior_rho_init=-3.0, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.prior_mean, self.prior_variance = prior_mean, prior_variance
        self.posterior_mu_init, self.posterior_rho_init = (posterior_mu_init,), (posterior_rho_init,)
        self.bias = bias
        mk = lambda n_in: LinearFlipout(in_features=n_in, out_features=4 * out_features, prior_mean=prior_mean,
                                     prior_variance=prior_variance, posterior_mu_init=posterior_mu_init,
                                     posterior_rho_init=posterior_rho_init, bias=bias)
        self.ih = mk(in_features)
        self.hh = mk(out_features)

    def kl_loss(self):
        return self.ih.kl_loss() + self.hh.kl_loss()

    def forward(self, X, hidden_states=None, return_kl=True):
        if self.dnn_to_bnn_flag:
            return_kl = False
        n, steps, _ = X.size()
        if hidden_states is None:
            h = torch.zeros(n, self.out_features, device=X.device)
            c = torch.zeros(n, self.out_features, device=X.device)
        else:
            h, c = hidden_states
        hs = self.out_features
        kl = 0
        hidden, cells = [], []
        for t in range(steps):
            a, kl_a = self.ih(X[:, t, :])
            b, kl_b = self.hh(h)
            gates = a + b
            kl = kl + kl_a + kl_b
            i_t, f_t = torch.sigmoid(gates[:, :hs]), torch.sigmoid(gates[:, hs:2 * hs])
            g_t, o_t = torch.tanh(gates[:, 2 * hs:3 * hs]), torch.sigmoid(gates[:, 3 * hs:])
            c = f_t * c + i_t * g_t
            h = o_t * torch.tanh(c)
            hidden.append(h)
            cells.append(c)
        hidden_seq = torch.stack(hidden, dim=1).contiguous()      # [batch, time, features]
        c_ts = torch.stack(cells, dim=1).contiguous()
        if return_kl:
            return hidden_seq, (hidden_seq, c_ts), kl
        return hidden_seq, (hidden_seq, c_ts)
