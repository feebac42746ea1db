// route.hip — key-range routing of k-min-mer occurrences for the multi-GPU path.
#include "mdbg_dev.h"
