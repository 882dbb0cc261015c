// nlohmann/json.hpp shim -- the image only has nlohmann-json 3.1.1 as a single header (/opt/conda/include/json.hpp),
// which predates basic_json::contains() (3.6.0) used by the reference's tests/test_keyswitch.cpp:70-73. For a key
// lookup on an object, contains(k) == (count(k) != 0), so the member name is mapped onto count(). Build-time only.
#pragma once
#include <fstream>
#include <memory>
#include <random>
#include <string>
#include <vector>
#include "/opt/conda/include/json.hpp"
#define contains count
