"""Metadata lives in pyproject.toml ([project], PEP 621).  setuptools < 61 (the 59.6 of the ROCm
image) cannot read it, so for those versions the same table is handed to setup() explicitly."""

import os

import setuptools

kwargs = {}
if int(setuptools.__version__.split(".")[0]) < 61:
    import tomli

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyproject.toml"), "rb") as fh:
        doc = tomli.load(fh)
    project, tool = doc["project"], doc["tool"]["setuptools"]
    kwargs = dict(
        name=project["name"],
        version=project["version"],
        description=project["description"],
        python_requires=project["requires-python"],
        install_requires=project["dependencies"],
        packages=tool["packages"],
        package_data=tool["package-data"],
        entry_points={"console_scripts": [f"{k} = {v}" for k, v in project["scripts"].items()]},
    )
setuptools.setup(**kwargs)
