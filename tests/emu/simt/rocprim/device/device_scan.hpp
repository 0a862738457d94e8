#pragma once
#include "../rocprim_fake.hpp"
