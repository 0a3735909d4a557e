// mock: nested name -> value list with get(name, default) writing the default back, as Teuchos does
#ifndef MOCK_TEUCHOS_PARAMETERLIST_HPP
#define MOCK_TEUCHOS_PARAMETERLIST_HPP
#include <any>
#include <map>
#include <memory>
#include <string>
namespace Teuchos {
class ParameterList {
 public:
  ParameterList() {}
  explicit ParameterList(const std::string& name) : name_(name) {}
  template <class T> ParameterList& set(const std::string& name, const T& value) { vals_[name] = value; return *this; }
  ParameterList& set(const std::string& name, const char* value) { vals_[name] = std::string(value); return *this; }
  template <class T> T& get(const std::string& name, T def_value) {
    auto it = vals_.find(name);
    if (it == vals_.end()) it = vals_.emplace(name, def_value).first;
    return std::any_cast<T&>(it->second);
  }
  std::string& get(const std::string& name, const char* def_value) { return get(name, std::string(def_value)); }
  bool isParameter(const std::string& name) const { return vals_.count(name) > 0; }
  bool isSublist(const std::string& name) const { return subs_.count(name) > 0; }
  ParameterList& sublist(const std::string& name) {
    auto it = subs_.find(name);
    if (it == subs_.end()) it = subs_.emplace(name, std::make_shared<ParameterList>(name)).first;
    return *it->second;
  }
  const std::string& name() const { return name_; }
 private:
  std::string name_;
  std::map<std::string, std::any> vals_;
  std::map<std::string, std::shared_ptr<ParameterList> > subs_;
};
}  // namespace Teuchos
#endif
