#include "kernels.h"
