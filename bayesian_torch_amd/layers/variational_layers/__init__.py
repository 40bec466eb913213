from .linear_variational import *
from .conv_variational import *
from .rnn_variational import *
