from .linear_variational import *
from .conv_variational import *
