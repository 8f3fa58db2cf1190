#include <string>
void rl_set_error(const std::string&) {}
