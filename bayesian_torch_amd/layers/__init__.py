from .flipout_layers import *
from .variational_layers import *
from .base_variational_layer import *
