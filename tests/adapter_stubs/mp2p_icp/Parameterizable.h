#pragma once
// stand-in: mp2p_icp_common Parameterizable.h:169-184 (DECLARE_PARAMETER_*: formula-capable parameters)
#include <mrpt/containers/yaml.h>
namespace mp2p_icp
{
class Parameterizable;
class ParameterSource  // Parameterizable.h:51-80
{
   public:
    void attach(Parameterizable& obj);
};
class Parameterizable
{
   public:
    virtual ~Parameterizable() = default;
    void         checkAllParametersAreRealized() const;
    virtual void attachToParameterSource(ParameterSource& source) { source.attach(*this); }  // :101
};
}  // namespace mp2p_icp
#define DECLARE_PARAMETER_REQ(Yaml__, Var__) Var__ = (Yaml__)[#Var__].as<decltype(Var__)>()
#define DECLARE_PARAMETER_OPT(Yaml__, Var__) Var__ = (Yaml__).getOrDefault<decltype(Var__)>(#Var__, Var__)
#define DECLARE_PARAMETER_IN_REQ(Yaml__, Var__, Parent__) Var__ = (Yaml__)[#Var__].as<decltype(Var__)>()
