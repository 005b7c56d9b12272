/* forwarding header: lets code written against the reference include "bpe.h" unchanged */
#pragma once
#include "../biogpt_compat.h"
