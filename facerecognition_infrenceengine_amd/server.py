"""The four HTTP routes of /root/reference/infrenceServer.py:685-724, byte-compatible JSON and
status codes, as an app factory with the managers injected (the reference builds module-level
singletons that connect to a database at import: :682-683)."""
from threading import Thread


def create_app(embedding_manager, camera_manager):
    from flask import Flask, jsonify, request
    app = Flask(__name__)

    @app.after_request
    def _cors(resp):                          # flask_cors.CORS(app) equivalent (infrenceServer.py:34)
        resp.headers["Access-Control-Allow-Origin"] = "*"
        resp.headers["Access-Control-Allow-Headers"] = "Content-Type"
        return resp

    @app.route("/api/embeddings/stats", methods=["GET"])
    def get_embedding_stats():
        return jsonify(embedding_manager.get_stats())

    @app.route("/api/embeddings/sync", methods=["POST"])
    def force_sync():
        try:
            embedding_manager.force_sync()
            return jsonify({"status": "success", "message": "Sync completed"})
        except Exception as e:
            return jsonify({"status": "error", "message": str(e)}), 500

    @app.route("/api/camera/start", methods=["POST"])
    def start_camera():
        data = request.json
        sources = data.get("sources", [0])
        company_id = data.get("company_id")
        if not company_id:
            return jsonify({"status": "error", "message": "Company ID required"}), 400
        try:
            Thread(target=camera_manager.start_cameras, args=(sources, company_id), daemon=True).start()
            return jsonify({"status": "success", "message": "Camera started"})
        except Exception as e:
            return jsonify({"status": "error", "message": str(e)}), 500

    @app.route("/api/camera/stop", methods=["POST"])
    def stop_camera():
        try:
            camera_manager.stop_cameras()
            return jsonify({"status": "success", "message": "Camera stopped"})
        except Exception as e:
            return jsonify({"status": "error", "message": str(e)}), 500

    return app
