"""YAML config object with the reference's surface (utils/YParams.py:4-54): attribute + item access, `in`,
`update_params`, `log`.  Loaded with PyYAML (ruamel is not required); YAML-1.2 style floats such as `1E-3` /
`1e-4` (config/swin.yaml:31,203), which PyYAML's YAML-1.1 resolver would return as strings, are resolved as floats.
"""
import logging
import re

import yaml


class _Loader(yaml.SafeLoader):
    pass


# YAML 1.2 core-schema float: optional sign, digits with optional fraction, optional exponent without a dot
_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(\.[0-9]+|[0-9]+(\.[0-9]*)?)([eE][-+]?[0-9]+)?$|^[-+]?\.(inf|Inf|INF)$|^\.(nan|NaN|NAN)$"),
    list("-+0123456789."))


def load_yaml(path):
    with open(path) as f:
        return yaml.load(f, Loader=_Loader)


class YParams():
    """ Yaml file parser """

    def __init__(self, yaml_filename, config_name, print_params=False):
        self._yaml_filename = yaml_filename
        self._config_name = config_name
        self.params = {}
        if print_params:
            print("------------------ Configuration ------------------")
        for key, val in load_yaml(yaml_filename)[config_name].items():
            if print_params:
                print(key, val)
            if isinstance(val, str) and val == 'None':
                val = None
            self.params[key] = val
            object.__setattr__(self, key, val)
        if print_params:
            print("---------------------------------------------------")

    def __setattr__(self, key, val):
        # both stores stay in sync (the reference swaps in a custom __setattr__ after construction)
        if not key.startswith('_') and key != 'params':
            self.__dict__.setdefault('params', {})[key] = val
        object.__setattr__(self, key, val)

    def __getitem__(self, key):
        return self.params[key]

    def __setitem__(self, key, val):
        self.params[key] = val
        object.__setattr__(self, key, val)

    def __contains__(self, key):
        return key in self.params

    def update_params(self, config):
        for key, val in config.items():
            self.params[key] = val
            object.__setattr__(self, key, val)

    def log(self):
        logging.info("------------------ Configuration ------------------")
        logging.info("Configuration file: " + str(self._yaml_filename))
        logging.info("Configuration name: " + str(self._config_name))
        for key, val in self.params.items():
            logging.info(str(key) + ' ' + str(val))
        logging.info("---------------------------------------------------")
