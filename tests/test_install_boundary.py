"""CPU: the zero-edit boundary (`wsmgmap.install()`, `python -m wsmgmap script.py`).

The reference's trainers import the policy and the auxiliary-loss registry by their reference names
(vlnce_baselines/common_trainer.py:24, vlnce_baselines/dagger_trainer.py:25).  After `install()` those names must resolve to the
product's class and to the ONE registry the product's policy registers into.  No reference file is used: the trainer side is a
package shell written into a temporary directory (empty `__init__.py` files and a module holding the two import statements)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ws-mgmap_amd")


def _shell(tmp_path):
    """A stand-in for the reference checkout: vlnce_baselines/{__init__,models/__init__}.py (as in the reference, `common` and
    `models/encoders` have no __init__) and a `trainer_shell` module that imports the two names the way the trainers do.  The
    shell's own policy.py / aux_losses.py raise: if an import ever reaches them, install() did not take effect."""
    vb = tmp_path / "vlnce_baselines"
    (vb / "models" / "encoders").mkdir(parents=True)
    (vb / "common").mkdir()
    (vb / "__init__.py").write_text("from vlnce_baselines import trainer_shell\n")     # the reference's __init__ imports its trainer
    (vb / "models" / "__init__.py").write_text("")
    (vb / "models" / "policy.py").write_text("raise ImportError('the reference policy file was reached')\n")
    (vb / "common" / "aux_losses.py").write_text("raise ImportError('the reference aux_losses file was reached')\n")
    (vb / "trainer_shell.py").write_text(textwrap.dedent("""
        from vlnce_baselines.models.policy import BasePolicy
        from vlnce_baselines.common.aux_losses import AuxLosses
    """))
    return tmp_path


def _run(code, cwd, extra_path=()):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([PKG, str(cwd), *extra_path, env.get("PYTHONPATH", "")])
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=str(cwd), env=env, capture_output=True, text=True, timeout=300)


def test_install_aliases_resolve_to_the_product_and_one_registry(tmp_path):
    _shell(tmp_path)
    r = _run("""
        import importlib, sys
        import wsmgmap
        names = wsmgmap.install()
        assert wsmgmap.installed() and "vlnce_baselines.models.policy" in names
        import vlnce_baselines                                    # runs the shell's __init__ -> trainer_shell's two imports
        from vlnce_baselines import trainer_shell
        import wsmgmap.models.policy as own
        import wsmgmap.common.aux_losses as own_aux
        assert importlib.import_module("vlnce_baselines.models.policy").BasePolicy is own.BasePolicy
        assert trainer_shell.BasePolicy is own.BasePolicy
        # ONE registry: the object the trainer activates / clears is the object the policy registers its losses into
        assert trainer_shell.AuxLosses is own_aux.AuxLosses is own.AuxLosses
        trainer_shell.AuxLosses.activate()
        assert own.AuxLosses.is_active()
        # the rest of the path's modules, and the `from package import module` spelling
        from vlnce_baselines.models import policy as p2
        assert p2 is own
        import vlnce_baselines.models.mg_map_policy as m
        assert m.MGMapNet is own.MGMapNet
        import vlnce_baselines.common.rgb_mapping as rm, wsmgmap.common.rgb_mapping as orm
        assert rm is orm
        import vlnce_baselines.models.encoders.map_encoder as me, wsmgmap.models.encoders.map_encoder as ome
        assert me.MapEncoder is ome.MapEncoder
        # a policy built through the reference name is the product's module tree
        from wsmgmap.config import default_model_config
        class Box: shape = (2,)
        pol = trainer_shell.BasePolicy(None, Box(), default_model_config(num_proc=1))
        assert type(pol.net).__module__ == "wsmgmap.models.mg_map_policy"
        assert wsmgmap.install() == names                         # idempotent
        wsmgmap.uninstall()
        assert "vlnce_baselines.models.policy" not in sys.modules and not wsmgmap.installed()
        print("ok")
    """, tmp_path)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_install_after_the_reference_modules_were_imported_is_refused(tmp_path):
    vb = tmp_path / "vlnce_baselines"
    (vb / "models").mkdir(parents=True)
    (vb / "__init__.py").write_text("")
    (vb / "models" / "__init__.py").write_text("")
    (vb / "models" / "policy.py").write_text("class BasePolicy: pass\n")
    r = _run("""
        import vlnce_baselines.models.policy as early
        import wsmgmap
        try:
            wsmgmap.install()
        except ImportError as e:
            assert "vlnce_baselines.models.policy" in str(e)
        else:
            raise SystemExit("install() after the fact must raise")
        wsmgmap.install(strict=False)
        import importlib, wsmgmap.models.policy as own
        assert importlib.import_module("vlnce_baselines.models.policy") is own
        import vlnce_baselines.models
        assert vlnce_baselines.models.policy is own               # the parent package's attribute follows
        print("ok")
    """, tmp_path)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_module_launcher_installs_before_the_script_runs(tmp_path):
    _shell(tmp_path)
    (tmp_path / "run_shell.py").write_text(textwrap.dedent("""
        import sys
        import vlnce_baselines
        from vlnce_baselines.trainer_shell import BasePolicy, AuxLosses
        import wsmgmap.models.policy as own
        assert BasePolicy is own.BasePolicy and AuxLosses is own.AuxLosses
        assert __name__ == "__main__" and sys.argv[1:] == ["--exp-config", "x.yaml"]
        print("launched ok")
    """))
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([PKG, env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, "-m", "wsmgmap", "run_shell.py", "--exp-config", "x.yaml"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "launched ok" in r.stdout, r.stdout + r.stderr


def test_install_without_any_reference_checkout_serves_package_shells(tmp_path):
    r = _run("""
        import importlib, sys
        import wsmgmap
        wsmgmap.install()
        import wsmgmap.models.policy as own
        assert importlib.import_module("vlnce_baselines.models.policy").BasePolicy is own.BasePolicy
        import vlnce_baselines.common.aux_losses as al
        assert al.AuxLosses is own.AuxLosses and sys.modules['vlnce_baselines'].common.aux_losses is al
        try:
            import vlnce_baselines.dagger_trainer       # NOT part of the path: nothing is invented for it
        except ModuleNotFoundError:
            pass
        else:
            raise SystemExit("only the path's modules may resolve")
        wsmgmap.uninstall()
        assert not [n for n in sys.modules if n.startswith("vlnce_baselines")]
        print("ok")
    """, tmp_path)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
