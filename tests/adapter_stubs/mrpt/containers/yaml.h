#pragma once
// stand-in: mrpt/containers/yaml.h (has / operator[] / as<T> and the MCP_LOAD_* macros)
#include <string>
namespace mrpt::containers
{
class yaml
{
   public:
    bool has(const std::string& key) const;
    yaml operator[](const std::string& key) const;
    template <class T> T as() const;
    template <class T> T getOrDefault(const std::string& key, const T& def) const;
    bool isSequence() const;
    bool isScalar() const;
    size_t size() const;
    yaml operator()(int index) const;
    yaml& operator[](const std::string& key);
    yaml& operator=(bool v);
};
}  // namespace mrpt::containers
#define MCP_LOAD_REQ(Yaml__, Var__) Var__ = (Yaml__)[#Var__].as<decltype(Var__)>()
#define MCP_LOAD_OPT(Yaml__, Var__) Var__ = (Yaml__).getOrDefault<decltype(Var__)>(#Var__, Var__)
