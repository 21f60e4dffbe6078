#pragma once
#include <string>
namespace mp2p_icp { using layer_name_t = std::string; }
