/* forwarding header: lets code written against the reference include "ggml.h" unchanged */
#pragma once
#include "../biogpt_compat.h"
