/* forwarding header: lets code written against the reference include "ggml-backend.h" unchanged */
#pragma once
#include "../biogpt_compat.h"
